"""BASELINE.json configs at their OWN shapes, and the collectives on the real backend.

* configs[1]: BM25 top-100, synthetic 100 k-doc postings resident in HBM, 1024 queries in ONE call (the L3-resident,
  49-tile regime that the 600 / 2 500 / 30 000 / 1 M-document tests only bracket), bitwise against the C oracle.
* a 1-rank RCCL smoke: torch.distributed backend "nccl" IS RCCL on ROCm.  world_size = 1 still loads librccl and runs the
  collectives of the sharded path in the forms it uses them (uint8 all_gather_into_tensor of the packed lists, also
  asynchronous and waited for later; all_reduce MIN of floats; int32 all_to_all_single with split sizes on slices of
  preallocated buffers; the int32 SUM all_reduce of raw cosine bits round 2 used), so the first 8-GPU run is not the
  first execution of those calls.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config2_bm25_100k_docs_1024_queries_top100_bitwise():
    assert torch.cuda.is_available()
    from msretr.engine import DeviceEngine
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    from oracle import bm25_ref, c_oracle
    ix = synthetic_corpus(100_000, n_chunks=0, n_terms=200_000, seed=11)
    terms, _ = synthetic_queries(ix, 1024, seed=12)
    eng = DeviceEngine(ix, max_queries=1024, max_k=100, rerank_max_docs=0)
    tl = [ix.term_ids(t) for t in terms]
    doc, score, n = [x.cpu().numpy() for x in eng.bm25_topk(tl, k=100)]      # ONE launch of 1024 queries
    z = {k: np.ascontiguousarray(getattr(ix, k).cpu().numpy()) for k in ("term_off", "post_doc", "post_tf", "doc_len", "idf")}
    z["avgdl"] = ix.avgdl
    rng = np.random.default_rng(5)
    sample = sorted(set(rng.choice(1024, size=48, replace=False).tolist()) | {0, 1023})
    for i in sample:
        ut, qtf = bm25_ref.prepare_query(tl[i], z["term_off"])
        oi, os_ = c_oracle.bm25_topk(z, ut, qtf, 100, 0.0, ix.k1, ix.b)
        assert n[i] == len(oi), i
        assert doc[i, :n[i]].tolist() == oi.tolist(), i
        assert score[i, :n[i]].tolist() == os_.tolist(), i               # float64, bitwise
        assert np.all(doc[i, n[i]:] == -1)
    # the same queries in slices of 100 (another grid shape) give the same bits
    eng2 = DeviceEngine(ix, max_queries=100, max_k=100, rerank_max_docs=0)
    d2, s2, n2 = [x.cpu().numpy() for x in eng2.bm25_topk(tl, k=100)]
    assert np.array_equal(d2, doc) and np.array_equal(s2, score) and np.array_equal(n2, n)
    eng.close(); eng2.close()


_RCCL_CHILD = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() %% 200))
    os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch, torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    from msretr.distributed import ShardedEngine
    se = ShardedEngine(engine=None, doc_base=0, row_base=0)
    se.world = 1                                        # helper under test takes its width from here
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    Q, k1, k2 = 5, 1000, 100
    parts = [torch.randint(-1, 10**6, (Q, k1), generator=g, dtype=torch.int32).to(dev),
             torch.randn((Q, k1), generator=g, dtype=torch.float64).to(dev),
             torch.randint(0, k1, (Q,), generator=g, dtype=torch.int32).to(dev),
             torch.randint(-1, 10**6, (Q, k2), generator=g, dtype=torch.int32).to(dev),
             torch.randn((Q, k2), generator=g).to(dev),
             torch.randint(-1, 5 * 10**6, (Q, k2), generator=g, dtype=torch.int32).to(dev),
             torch.randint(0, k2, (Q,), generator=g, dtype=torch.int32).to(dev)]
    got = se._allgather_bytes(parts)
    ok_gather = len(got) == 1 and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(got[0], parts))
    cos = torch.randn((Q, k1, 10), generator=g).to(dev)
    cos[0, 0, 0] = -0.0                                 # raw bits must survive: -0.0 stays -0.0
    meta = torch.randint(0, 2**30, (Q, k1, 3), generator=g, dtype=torch.int32).to(dev)
    buf = torch.cat([cos.view(torch.int32).reshape(-1), meta.reshape(-1)])
    ref = buf.clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    ok_reduce = torch.equal(buf, ref)
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
    # the calls of ShardedEngine.search in the forms it uses them: an asynchronous all-gather waited for later, an
    # all-reduce MIN of floats, an all-to-all with split sizes on slices of preallocated int32 buffers
    src = torch.arange(4096, dtype=torch.uint8, device=dev); dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(dst, src, async_op=True)
    mn = torch.tensor([0.5, -1.0], device=dev); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    work.wait()
    send = torch.arange(64 * 16, dtype=torch.int32, device=dev); recv = torch.zeros(80 * 16, dtype=torch.int32, device=dev)
    dist.all_to_all_single(recv[:37 * 16], send[:37 * 16], [37 * 16], [37 * 16])
    torch.cuda.synchronize()
    ok_forms = torch.equal(dst, src) and mn.tolist() == [0.5, -1.0] and torch.equal(recv[:37 * 16], send[:37 * 16]) and int(recv[37 * 16:].abs().sum()) == 0
    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l or "libnccl" in l]
    print(json.dumps({"gather": bool(ok_gather), "reduce": bool(ok_reduce), "max": float(t.item()), "forms": bool(ok_forms),
                      "backend": dist.get_backend(), "lib": sorted(set(loaded))[:2]}))
    dist.destroy_process_group()
""")


def test_rccl_one_rank_smoke():
    assert torch.cuda.is_available()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["gather"] and out["reduce"] and out["max"] == 1.25 and out["forms"]
    assert out["lib"], "librccl was not mapped into the process"
