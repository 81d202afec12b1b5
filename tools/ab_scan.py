#!/usr/bin/env python3
"""A/B of the dense-scan kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
Kernel time comes from the engine's hipEvent pairs on the launch stream.

    python tools/ab_scan.py [--chunks 5000000] [--docs 1000000] [--rounds 7]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from msretr import _abi  # noqa: E402
from msretr.build import build_library  # noqa: E402
from msretr.engine import DeviceEngine  # noqa: E402
from msretr.synthetic import synthetic_corpus  # noqa: E402

if os.environ.get("MSR_DIAG_LIB"):       # A/B knobs (MSR_SCAN_DEBUG, MSR_KS_PIPE, ...) only exist in the -DMSR_DIAG build
    _abi.LIB_PATH = build_library(diag=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--chunks", type=int, default=5_000_000)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--configs", default="0:14,0:7,0:2,0:15", help="layout:variant list")
    ap.add_argument("--queries", default="1,16,32")
    ap.add_argument("--nonzero", type=int, default=-1,
                    help="diagnostic: keep only the first N query rows non-zero (the MFMA count stays, operand data changes)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ix = synthetic_corpus(args.docs, n_chunks=args.chunks, device=dev, with_postings=False)
    torch.cuda.synchronize()
    cfgs = [tuple(int(x) for x in c.split(":")) for c in args.configs.split(",")]
    engines = {c: DeviceEngine(ix, max_queries=32, max_k=100, rerank_max_docs=0, scan_layout=c[0], scan_variant=c[1]) for c in cfgs}
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    qs = {int(q): torch.randn((int(q), 768), generator=g).to(dev) for q in args.queries.split(",")}
    if args.nonzero >= 0:
        for v in qs.values():
            v[args.nonzero:] = 0
    alg = args.chunks * 768 * 4
    res = {}
    ref = {}
    for rnd in range(args.rounds + 1):
        for c, e in engines.items():
            for Q, qv in qs.items():
                e.set_timing(True)
                out = e.dense_topk(qv, k=100, want_chunk=False)
                torch.cuda.synchronize()
                ms, n = e.kernel_time_ms(0)
                e.set_timing(False)
                if rnd:                                   # round 0 = warm-up
                    res.setdefault((c, Q), []).append(ms / n)
                key = Q
                if args.nonzero >= 0:
                    continue                              # diagnostic run: results are degenerate, timing only
                if key not in ref:
                    ref[key] = [x.cpu() for x in (out[0], out[1])]
                else:
                    # kernels differ in summation order: same scores within rounding, same documents up to near-ties
                    assert torch.allclose(out[1].cpu(), ref[key][1], atol=2e-6), ("scores differ", c, Q)
                    assert float((out[0].cpu() == ref[key][0]).float().mean()) > 0.995, ("doc ids differ between variants", c, Q)
    rows = []
    for (c, Q), v in sorted(res.items()):
        med, mn = float(np.median(v)), float(np.min(v))
        rows.append({"layout": c[0], "variant": c[1], "queries": Q, "median_ms": med, "min_ms": mn,
                     "GBps_median": alg / med / 1e6, "frac_of_8TBps": alg / med / 1e6 / 8000})
        print(f"layout {c[0]} variant {c[1]} Q={Q:2d}: median {med:.3f} ms  min {mn:.3f} ms  {alg / med / 1e6:7.0f} GB/s "
              f"({100 * alg / med / 1e6 / 8000:.1f} % of 8 TB/s)", flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
