#!/usr/bin/env python3
"""The BM25 stage's select and the merge of the shards' lists on their own (K4, `msr_topk.hip`), timed with events; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.  MSR_LIB_PATH=<other build> for an A/B on one box.
    python tools/select_bench.py [--docs 1000000] [--queries 256] [--shards 8] [--iters 20]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msretr import _abi  # noqa: E402

if os.environ.get("MSR_LIB_PATH"):
    _abi.LIB_PATH = os.environ["MSR_LIB_PATH"]
from msretr.engine import DeviceEngine  # noqa: E402
from msretr.synthetic import SEED, synthetic_corpus, synthetic_queries  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=1_000_000)
ap.add_argument("--terms", type=int, default=200_000)
ap.add_argument("--queries", type=int, default=256)
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--k", type=int, default=1000)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


full = synthetic_corpus(a.docs, n_chunks=0, n_terms=a.terms, seed=SEED, device=dev)
out = {"lib": os.path.basename(_abi.LIB_PATH)}
for world, nq in ((1, a.queries), (a.shards, a.queries * a.shards)):
    terms, _ = synthetic_queries(full, nq, seed=777, device="cpu")
    shard = full.shard(0, world) if world > 1 else full
    e = DeviceEngine(shard, max_queries=nq, max_k=a.k, rerank_max_docs=0)
    packed = e.pack_queries(terms)
    res = e.bm25_topk(None, k=a.k, packed=packed)
    out[f"bm25_topk_ms_{world}shard_{nq}q"] = timed(lambda: e.bm25_topk(None, k=a.k, packed=packed), a.iters)
    out[f"n_mean_{world}shard"] = float(res[2].float().mean())
    if world > 1:
        # the merge a rank runs after the all-gather: its nq / world own queries, `world` sorted lists each
        Q = nq // world
        g = torch.Generator(device="cpu"); g.manual_seed(3)
        sc = torch.rand((world, Q, a.k), generator=g, dtype=torch.float64).sort(dim=2, descending=True).values.to(dev)
        docs = torch.stack([torch.randperm(a.docs, generator=g)[:Q * a.k].reshape(Q, a.k) for _ in range(world)]).to(torch.int32).to(dev)
        ns = torch.full((world, Q), a.k, dtype=torch.int32, device=dev)
        m = e.merge_topk(docs, sc, ns, a.k)
        ref = torch.cat([sc[p] for p in range(world)], dim=1).sort(dim=1, descending=True).values[:, :a.k]
        out["merge_ok"] = bool(torch.equal(m[1], ref))
        out[f"merge_ms_{world}x{a.k}_{Q}q"] = timed(lambda: e.merge_topk(docs, sc, ns, a.k), a.iters)
        scf = sc.float().sort(dim=2, descending=True).values[:, :, :100].contiguous()
        out[f"merge_f32_ms_{world}x100_{Q}q"] = timed(lambda: e.merge_topk(docs[:, :, :100].contiguous(), scf, torch.full_like(ns, 100), 100), a.iters)
    e.close()
print(json.dumps(out), flush=True)
