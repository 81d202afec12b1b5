#!/usr/bin/env python3
"""Kernel time of the dense sweeps (msr_dense_topk with <= 64 queries: the K-split kernel; the single-query latency path) at
the benchmark's size, product or diagnostic library (MSR_DIAG_LIB=1).
    python tools/sweep_time.py [--queries 1 32 64]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msretr import _abi  # noqa: E402
from msretr.build import build_library  # noqa: E402
from msretr.engine import DeviceEngine  # noqa: E402
from msretr.synthetic import synthetic_corpus  # noqa: E402

if os.environ.get("MSR_DIAG_LIB"):
    _abi.LIB_PATH = build_library(diag=True)
ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, nargs="+", default=[1, 32, 64])
ap.add_argument("--chunks", type=int, default=5_000_000)
ap.add_argument("--docs", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=9)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ix = synthetic_corpus(a.docs, n_chunks=a.chunks, device=dev, with_postings=False)
e = DeviceEngine(ix, max_queries=64, max_k=100, rerank_max_docs=0)
g = torch.Generator(device="cpu"); g.manual_seed(5)
for Q in a.queries:
    q = torch.randn((Q, 768), generator=g).to(dev)
    ts = []
    for it in range(a.iters + 2):
        e.set_timing(True)
        e.dense_topk(q, k=100)
        torch.cuda.synchronize()
        ms, n = e.kernel_time_ms(0)
        e.set_timing(False)
        if it >= 2:
            ts.append(ms / max(1, n))
    ts.sort()
    med = ts[len(ts) // 2]
    print(json.dumps({"library": os.path.basename(_abi.LIB_PATH), "queries": Q, "sweep_kernel_ms_median": med, "min": ts[0],
                      "TBps": a.chunks * 768 * 4 / (med * 1e-3) / 1e12}), flush=True)
e.close()
