#!/usr/bin/env python3
"""BM25 stage (msr_bm25_topk) timings on a synthetic corpus, with the diagnostic build's knock-out switches.
    MSR_DIAG_LIB=1 python tools/bm25_check.py --docs 1000000 --queries 128 --k 1000 --dbg 0,1,2,4,16"""
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

if os.environ.get("MSR_LIB"):            # a variant library built by hand (timing experiments)
    _abi.LIB_PATH = os.environ["MSR_LIB"]
elif os.environ.get("MSR_DIAG_LIB"):       # timing experiments (--dbg) only exist in the -DMSR_DIAG build
    _abi.LIB_PATH = build_library(diag=True)
from msretr.synthetic import synthetic_corpus, synthetic_queries  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=1_000_000)
ap.add_argument("--terms", type=int, default=1_000_000)
ap.add_argument("--queries", type=int, default=128)
ap.add_argument("--k", type=int, default=1000)
ap.add_argument("--iters", type=int, default=7)
ap.add_argument("--dbg", default="0", help="comma-separated knock-out masks (msr_tune(102, mask)); non-zero: diagnostic build only")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ix = synthetic_corpus(a.docs, n_chunks=0, n_terms=a.terms, device=dev)
terms, _ = synthetic_queries(ix, a.queries, seed=6)
e = DeviceEngine(ix, max_queries=a.queries, max_k=a.k, rerank_max_docs=0)
toff = ix.term_off
tq = torch.tensor([t for q in terms for t in set(q)], device=toff.device)
n_post = int((toff[tq + 1] - toff[tq]).sum())
for dbg in [int(x) for x in a.dbg.split(",")]:
    if dbg or os.environ.get("MSR_DIAG_LIB"):
        e._check(e.lib.msr_tune(e.handle, 102, dbg))
    ts = []
    for it in range(a.iters):
        e.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = e.bm25_topk(terms, k=a.k)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ms, n = e.kernel_time_ms(1)
        e.set_timing(False)
        ts.append((1e3 * (t1 - t0), ms / max(1, n)))
    tot, km = [sorted(x)[len(x) // 2] for x in zip(*ts)]
    print(json.dumps({"dbg": dbg, "queries": a.queries, "postings_per_launch": n_post, "stage_ms": tot, "full_pass_kernel_ms": km,
                      "GBps_postings": 8 * n_post / (km * 1e-3) / 1e9, "returned": int(out[2].sum())}), flush=True)
e.close()
