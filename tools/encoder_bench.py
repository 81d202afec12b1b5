#!/usr/bin/env python3
"""Throughput of the query encoder (msretr.encoder.QueryEncoder, random ModernBERT-base weights) next to transformers'
eager ModernBertModel on the same GPU and weights.

    python tools/encoder_bench.py [--queries 128] [--tokens 8] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from msretr.encoder import QueryEncoder  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=128)
ap.add_argument("--tokens", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--hip-linear-max", type=int, default=-1,
                help="tokens up to which the projections run on msr_enc_linear; beyond: the library GEMM, for comparison "
                     "(default: msr_enc_linear always)")
ap.add_argument("--no-hf", action="store_true", help="skip the transformers comparison (profiling runs)")
a = ap.parse_args()
if a.hip_linear_max >= 0:
    # comparison runs only: batches of more than --hip-linear-max tokens take the library GEMM (hipBLASLt through torch) instead
    # of msr_enc_linear.  The product module has no such branch; it is patched in here.
    _hip_linear = QueryEncoder._linear

    def _linear_or_library(self, x, weight, y, resid=None):
        if x.shape[0] <= a.hip_linear_max:
            return _hip_linear(self, x, weight, y, resid)
        if resid is None:
            torch.mm(x, weight.t(), out=y)
        elif resid is y:
            y.addmm_(x, weight.t())
        else:
            torch.addmm(resid, x, weight.t(), out=y)
        return y

    QueryEncoder._linear = _linear_or_library
from transformers import ModernBertConfig, ModernBertModel  # noqa: E402
torch.manual_seed(0)
hf = ModernBertModel(ModernBertConfig(reference_compile=False, attn_implementation="eager")).eval().cuda()
enc = QueryEncoder(hf.state_dict(), device=0)
rng = np.random.default_rng(1)
res = {"queries": a.queries, "tokens_per_query": a.tokens}
for nq in (1, a.queries):
    seqs = [rng.integers(0, 50000, size=a.tokens).tolist() for _ in range(nq)]
    ids = torch.tensor(seqs).cuda()
    mask = torch.ones_like(ids)

    def ours():
        return enc.encode(seqs)

    def theirs():
        with torch.no_grad():
            return hf(input_ids=ids, attention_mask=mask).last_hidden_state.mean(1)

    def ours_eager():
        enc.use_graphs = False
        try:
            return enc.encode(seqs)
        finally:
            enc.use_graphs = True

    for name, fn in (("msretr_hipgraph", ours), ("msretr_launches", ours_eager)) + (() if a.no_hf else (("transformers_eager", theirs),)):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = fn()
        torch.cuda.synchronize()
        res[f"{name}_ms_per_batch_of_{nq}"] = round(1e3 * (time.perf_counter() - t0) / a.iters, 3)
    if not a.no_hf:
        res[f"max_abs_diff_batch_of_{nq}"] = float((ours() - theirs()).abs().max())
print(json.dumps(res))
