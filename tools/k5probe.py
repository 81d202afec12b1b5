#!/usr/bin/env python3
"""Batched candidate path (msr_dense_topk_bf16, > 128 queries) against 64-query calls of the exact scan on a small corpus, with
planted near-duplicate queries -- the quick correctness probe used while the streaming kernel was brought up on bf16 rows.
    [MSR_DIAG_LIB=1] python tools/k5probe.py <queries> <docs> <chunks> [dbg]      (dbg: msr_tune(100, dbg), diagnostic build only)"""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msretr import _abi
from msretr.build import build_library
if os.environ.get("MSR_DIAG_LIB"):
    _abi.LIB_PATH = build_library(diag=True)
from msretr.engine import DeviceEngine
from msretr.synthetic import synthetic_corpus
nq = int(sys.argv[1]); docs = int(sys.argv[2]); chunks = int(sys.argv[3])
dev = torch.device("cuda", 0)
ix = synthetic_corpus(docs, n_chunks=chunks, device=dev, with_postings=False)
e = DeviceEngine(ix, max_queries=64, max_k=100, rerank_max_docs=0)
e.enable_bf16()
if len(sys.argv) > 4:
    e._check(e.lib.msr_tune(e.handle, 100, int(sys.argv[4])))
g = torch.Generator(device="cpu"); g.manual_seed(5)
rows = torch.randint(0, chunks, (nq,), generator=g)
q = (ix.emb[rows.to(dev)] * 3.0 + 0.3 * torch.randn((nq, 768), generator=g).to(dev))
out = e.dense_topk_batched(q, k=100)
torch.cuda.synchronize()
ex = torch.cat([torch.stack(e.dense_topk(q[s:s + 64], k=100)[:2]) for s in range(0, nq, 64)], dim=1)
bad = ((out[0] != ex[0].to(torch.int32)) & ((out[1] - ex[1]).abs() > 2e-6)).any(dim=1)
print("ok", nq, docs, chunks, "max diff", float((out[1] - ex[1]).abs().max()), "top1 equal", float((out[0][:, 0] == ex[0][:, 0].to(torch.int32)).float().mean()),
      "bad queries", bad.nonzero().flatten().tolist()[:40], "n_bad", int(bad.sum()))
