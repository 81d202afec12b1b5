#!/usr/bin/env python3
"""Run the dense scan a few times in one configuration (for rocprofv3 --pmc passes).
    python tools/scan_once.py --variant 0 --layout 0 --queries 32 --iters 3"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msretr import _abi  # noqa: E402
from msretr.build import build_library  # noqa: E402
from msretr.engine import DeviceEngine  # noqa: E402
from msretr.synthetic import synthetic_corpus  # noqa: E402

if os.environ.get("MSR_DIAG_LIB"):       # the MSR_* knobs only exist in the -DMSR_DIAG build
    _abi.LIB_PATH = build_library(diag=True)

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--layout", type=int, default=0)
ap.add_argument("--queries", type=int, default=32)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--chunks", type=int, default=5_000_000)
ap.add_argument("--docs", type=int, default=1_000_000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ix = synthetic_corpus(a.docs, n_chunks=a.chunks, device=dev, with_postings=False)
e = DeviceEngine(ix, max_queries=32, max_k=100, rerank_max_docs=0, scan_layout=a.layout, scan_variant=a.variant)
g = torch.Generator(device="cpu"); g.manual_seed(5)
q = torch.randn((a.queries, 768), generator=g).to(dev)
for _ in range(a.iters):
    e.dense_topk(q, k=100, want_chunk=False)
torch.cuda.synchronize()
e.close()
