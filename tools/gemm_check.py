#!/usr/bin/env python3
"""GEMM candidate path (msr_dense_topk_bf16 with > 128 queries) against the exact f32 scan, with timings.
    python tools/gemm_check.py --docs 200000 --chunks 1000000 --queries 1024 --k 100 [--iters 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msretr import _abi  # noqa: E402
from msretr.build import build_library  # noqa: E402
from msretr.engine import DeviceEngine  # noqa: E402

if os.environ.get("MSR_DIAG_LIB"):       # timing experiments (--dbg) only exist in the -DMSR_DIAG build
    _abi.LIB_PATH = build_library(diag=True)
if os.environ.get("MSR_LIB_PATH"):       # A/B against another build of the library (e.g. the previous commit's)
    _abi.LIB_PATH = os.environ["MSR_LIB_PATH"]
from msretr.synthetic import synthetic_corpus  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=200_000)
ap.add_argument("--chunks", type=int, default=1_000_000)
ap.add_argument("--queries", type=int, default=1024)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--check", type=int, default=128, help="queries compared with the exact path")
ap.add_argument("--dbg", type=int, default=0, help="diagnostic build only: msr_tune(100, dbg) before the timing loops")
ap.add_argument("--dbg-stream", type=int, default=0, help="diagnostic build only: msr_tune(101, v), the streaming kernel's switches")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ix = synthetic_corpus(a.docs, n_chunks=a.chunks, device=dev, with_postings=False)
e = DeviceEngine(ix, max_queries=64, max_k=max(a.k, 100), rerank_max_docs=0)
e.enable_bf16()
print("gemm_ok", e.batch_gemm_ok(), "batch_width", e.batch_width(), flush=True)
g = torch.Generator(device="cpu"); g.manual_seed(5)
rows = torch.randint(0, a.chunks, (a.queries,), generator=g)
q = (ix.emb[rows.to(dev)] + 0.5 * torch.nn.functional.normalize(torch.randn((a.queries, 768), generator=g), dim=1).to(dev)) * 7.0
out = e.dense_topk_batched(q, k=a.k)
torch.cuda.synchronize()
n_bad = int((out[3] < a.k).sum())
ex = e.dense_topk(q[:a.check], k=a.k)
torch.cuda.synchronize()
d_err = float((out[1][:a.check] - ex[1]).abs().max())
agree = float((out[0][:a.check] == ex[0]).float().mean())
tie_ok = bool(((out[0][:a.check] == ex[0]) | ((out[1][:a.check] - ex[1]).abs() <= 2e-6)).all())
print(json.dumps({"short_rows": n_bad, "max_abs_score_diff": d_err, "doc_agreement": agree, "near_tie_only": tie_ok}), flush=True)
if a.dbg:
    e._check(e.lib.msr_tune(e.handle, 100, a.dbg))
if a.dbg_stream:
    e._check(e.lib.msr_tune(e.handle, 101, a.dbg_stream))
for ver in (3,):
  o2 = e.dense_topk_batched(q, k=a.k)
  torch.cuda.synchronize()
  same = all(torch.equal(x, y) for x, y in zip(o2, out))
  ts = []
  for it in range(a.iters):
    e.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.dense_topk_batched(q, k=a.k)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    emit_ms, n2 = e.kernel_time_ms(2)
    samp_ms, n3 = e.kernel_time_ms(3)
    e.set_timing(False)
    ts.append((1e3 * (t1 - t0), emit_ms / max(1, n2), samp_ms / max(1, n3)))
  tot, em, sm = [sorted(x)[len(x) // 2] for x in zip(*ts)]
  flops = 2.0 * 768 * a.chunks * ((a.queries + 255) // 256 * 256)
  print(json.dumps({"version": ver, "same_as_first_run": same, "queries": a.queries, "total_ms": tot, "emit_pass_ms": em,
                    "sample_pass_ms": sm, "emit_TFLOPs": flops / (em * 1e-3) / 1e12, "qps": a.queries / (tot * 1e-3)}), flush=True)
e.close()
