#!/usr/bin/env python3
"""msr_enc_linear on the four projection shapes of a ModernBERT layer for a BATCH of queries, next to the library GEMM
(torch.mm = hipBLASLt) on the same operands; with MSR_DIAG_LIB=1 every tile shape of the kernel is timed.
    MSR_DIAG_LIB=1 python tools/enc_linear_bench.py --tokens 1024"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msretr import _abi  # noqa: E402
from msretr.build import build_library  # noqa: E402

diag = bool(os.environ.get("MSR_DIAG_LIB"))
if diag:
    _abi.LIB_PATH = build_library(diag=True)
lib = _abi.load()
ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=1024)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / a.iters


out = []
for n_out, n_in in ((2304, 768), (768, 768), (768, 1152)):
    x = torch.randn(a.tokens, n_in, device=dev)
    w = torch.randn(n_out, n_in, device=dev) * 0.05
    y = torch.empty(a.tokens, n_out, device=dev)
    flops = 2.0 * a.tokens * n_out * n_in
    row = {"n_out": n_out, "n_in": n_in, "tokens": a.tokens}
    row["library_us"] = timed(lambda: torch.mm(x, w.t(), out=y))
    for code in ([0, 1, 2, 5, 1 + 16 * 30, 1 + 16 * 40, 2 + 16 * 70] if diag else [0]):
        if diag:
            lib.msr_enc_linear_force_shape(code)
        us = timed(lambda: lib.msr_enc_linear(P(x), P(w), None, P(y), a.tokens, n_out, n_in, S))
        row[f"hip_shape{code}_us"] = us
        row[f"hip_shape{code}_TFLOPs"] = flops / us / 1e6
    row["library_TFLOPs"] = flops / row["library_us"] / 1e6
    out.append(row)
    print(json.dumps(row), flush=True)
