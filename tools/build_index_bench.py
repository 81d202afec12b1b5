#!/usr/bin/env python3
"""Time the sort-based BM25 table build (msretr.index_build.bm25_index_from_token_ids) on the GPU and on the CPU device.

    python tools/build_index_bench.py [--docs 200000] [--mean-len 370] [--terms 200000]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from msretr.index_build import bm25_index_from_token_ids  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=200_000)
ap.add_argument("--mean-len", type=int, default=370)
ap.add_argument("--terms", type=int, default=200_000)
a = ap.parse_args()
rng = np.random.default_rng(3)
lens = np.clip(rng.lognormal(np.log(a.mean_len) - 0.32, 0.8, size=a.docs), 8, 20000).astype(np.int64)
off = np.zeros(a.docs + 1, np.int64); off[1:] = np.cumsum(lens)
tok = (rng.zipf(1.07, size=int(off[-1])) % a.terms).astype(np.int32)
ids = np.arange(a.docs, dtype=np.int64) * 2 + 1
res = {"docs": a.docs, "tokens": int(off[-1]), "terms": a.terms}
for dev in ("cuda", "cuda", "cpu"):                       # first cuda pass = warm-up
    t0 = time.time()
    ix = bm25_index_from_token_ids(ids, off, tok, a.terms, device=dev)
    if dev == "cuda":
        torch.cuda.synchronize()
    res[dev + "_seconds"] = round(time.time() - t0, 3)
    res["postings"] = int(ix.post_doc.numel())
print(json.dumps(res))
