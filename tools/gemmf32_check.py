#!/usr/bin/env python3
"""The 128-query GEMM scan over the f32 rows (msr_dense_topk with more than 64 queries) against the 64-query sweeps.
    python tools/gemmf32_check.py --docs 1000000 --chunks 5000000 [--iters 5]"""
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
ap.add_argument("--queries", type=int, default=128)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dbg", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ix = synthetic_corpus(a.docs, n_chunks=a.chunks, device=dev, with_postings=False)
e = DeviceEngine(ix, max_queries=max(64, a.queries), max_k=max(a.k, 100), rerank_max_docs=0)
print("scan_width", e.scan_width(), "arith", e.scan_arith(), flush=True)
g = torch.Generator(device="cpu"); g.manual_seed(5)
rows = torch.randint(0, a.chunks, (a.queries,), generator=g)
q = (ix.emb[rows.to(dev)] + 0.5 * torch.nn.functional.normalize(torch.randn((a.queries, 768), generator=g), dim=1).to(dev)) * 7.0
got = e.dense_topk(q, k=a.k)
torch.cuda.synchronize()
ref = [torch.cat(x) for x in zip(*[e.dense_topk(q[s:s + 64], k=a.k) for s in range(0, a.queries, 64)])]
torch.cuda.synchronize()
d_err = float((got[1] - ref[1]).abs().max())
same = got[0] == ref[0]
print(json.dumps({"n_equal": bool(torch.equal(got[3], ref[3])), "max_abs_score_diff": d_err, "doc_agreement": float(same.float().mean()),
                  "near_tie_only": bool((same | ((got[1] - ref[1]).abs() <= 2e-6)).all()),
                  "chunk_agreement": float(((got[2] == ref[2]) | ~same).float().mean())}), flush=True)
if a.dbg:
    e._check(e.lib.msr_tune(e.handle, 101, a.dbg))
print("dense_path", e.dense_path(), flush=True)
for name, fn in (("stream_pass", lambda: e.dense_topk(q, k=a.k)),
                 ("sweeps_2x64", lambda: [e.dense_topk(q[s:s + 64], k=a.k) for s in range(0, a.queries, 64)])):
    ts = []
    for it in range(a.iters):
        e.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ms, n = e.kernel_time_ms(0)
        e.set_timing(False)
        ts.append((1e3 * (t1 - t0), ms / max(1, n), n))
    tot, km, n = sorted(ts)[len(ts) // 2]
    print(json.dumps({"path": name, "total_ms": tot, "scan_kernel_ms_per_launch": km, "launches": n,
                      "GBps": a.chunks * 768 * 4 / (km * 1e-3) / 1e9}), flush=True)
e.close()
