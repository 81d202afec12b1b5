"""N > 1 path on CPU: two (and three) gloo ranks, each with its document shard, must reproduce the unsharded result
exactly (bm25 + dense lists after their all-gathers + merges; rerank after the all-to-all of the owned slots' records, and in
its dense form).  The compute is the oracle (tests/oracle_engine.py); what is under test is msretr.distributed +
CorpusIndex.shard."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus():
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    ix = synthetic_corpus(900, n_chunks=4200, n_terms=700, seed=21)
    terms, qvec = synthetic_queries(ix, 5, seed=22)
    # make some documents share a URL (dedup must work across shards) and drop one from urlsDB
    urls = [f"https://h{d % 37}.de/p{d}" for d in range(900)]
    urls[10] = urls[700].split("?")[0] + "?x=1"
    urls[700] = urls[700]
    urls[450] = None
    ix.urls = urls
    return ix, terms, qvec


def _run(rank, world, port, ret, snap_dir=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from msretr.distributed import ShardedEngine
        from msretr.index import CorpusIndex
        from oracle_engine import OracleEngine
        ix, terms, qvec = _corpus()
        sh = ix.shard(rank, world)
        if snap_dir:                                  # every rank serves its shard from a snapshot it reloaded
            d = os.path.join(snap_dir, f"shard{rank}")
            sh.save_dir(d)
            sh = CorpusIndex.load_dir(d, mmap=False)
        se = ShardedEngine(OracleEngine(sh), sh.doc_base, sh.row_base)
        assert se.world == world and se.rank == rank
        out = se.search([sh.term_ids(t) for t in terms], qvec, k1=200, k2=50)
        assert se.engine.begin_calls == 1                  # the dense stage went through begin / all-reduce MIN / end
        assert se._bounds[0] is not None                   # ... and the rerank exchange took its compact form (records)
        res = {k: [x.numpy() for x in v] for k, v in out.items()}
        dense_form = ShardedEngine(se.engine, sh.doc_base, sh.row_base, a2a="blocks")
        res["rerank_blocks"] = [x.numpy() for x in dense_form.search([sh.term_ids(t) for t in terms], qvec, k1=200, k2=50)["rerank"]]
        cut = se.search([sh.term_ids(t) for t in terms], qvec, k1=200, k2=50, rerank_keep=20)
        res["rerank_cut"] = [x.numpy() for x in cut["rerank"]]
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("via_snapshot", [False, True])
def test_two_rank_sharded_equals_unsharded(via_snapshot, tmp_path):
    """via_snapshot: the shards go through save_dir / load_dir first -- the URL groups must stay the corpus-wide ones
    (a shard that renumbered them from its own URLs would dedup unrelated documents of different shards against each
    other: round-1 advisor finding)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from msretr.distributed import ShardedEngine
    from oracle_engine import OracleEngine
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000 + (7 if via_snapshot else 0)
    mp.spawn(_run, args=(world, port, ret, str(tmp_path) if via_snapshot else None), nprocs=world, join=True)
    ix, terms, qvec = _corpus()
    ref = ShardedEngine(OracleEngine(ix), 0, 0).search([ix.term_ids(t) for t in terms], qvec, k1=200, k2=50)
    ref = {k: [x.numpy() for x in v] for k, v in ref.items()}
    for r in range(world):
        got = ret[r]
        for key in ("bm25", "dense", "rerank", "rerank_blocks"):            # (rerank_blocks: the dense form of the exchange)
            for a, b in zip(got[key], ref[key.replace("_blocks", "")]):
                assert a.shape == b.shape and np.array_equal(a, b), (r, key)
        # rerank_keep < k1: lists AND counts are cut (a caller iterating range(n[q]) stays inside the rows it was given)
        cut = got["rerank_cut"]
        assert cut[0].shape == (len(terms), 20) and int(cut[4].max()) <= 20
        assert np.array_equal(cut[4], np.minimum(ref["rerank"][4], 20)) and np.array_equal(cut[5], ref["rerank"][5])
        for j in range(4):
            assert np.array_equal(cut[j], ref["rerank"][j][:, :20])
    # both ranks hold identical results
    for key in ("bm25", "dense", "rerank"):
        for a, b in zip(ret[0][key], ret[1][key]):
            assert np.array_equal(a, b)


def _run_few(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from msretr.distributed import ShardedEngine
        from oracle_engine import OracleEngine
        ix, terms, qvec = _corpus()
        sh = ix.shard(rank, world)
        se = ShardedEngine(OracleEngine(sh), sh.doc_base, sh.row_base)
        res = {}
        for nq in (2, 4):                                   # 2 queries on 3 ranks: the last rank fuses none; 4: a ragged last block
            out = se.search([sh.term_ids(t) for t in terms[:nq]], qvec[:nq], k1=120, k2=30)
            res[nq] = {k: [x.numpy() for x in v] for k, v in out.items()}
        assert se._bounds[0] is not None
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_ranks_with_fewer_queries_than_ranks_and_a_ragged_block():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from msretr.distributed import ShardedEngine
    from oracle_engine import OracleEngine
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_few, args=(world, 30000 + os.getpid() % 900, ret), nprocs=world, join=True)
    ix, terms, qvec = _corpus()
    for nq in (2, 4):
        ref = ShardedEngine(OracleEngine(ix), 0, 0).search([ix.term_ids(t) for t in terms[:nq]], qvec[:nq], k1=120, k2=30)
        ref = {k: [x.numpy() for x in v] for k, v in ref.items()}
        for r in range(world):
            for key in ("bm25", "dense", "rerank"):
                for j, (a, b) in enumerate(zip(ret[r][nq][key], ref[key])):
                    if key == "dense" and j == 1:            # (the stand-in's scores come out of a BLAS product whose blocking
                        #                                      follows the matrix shape: a shard's differ from the whole corpus' in the last bit)
                        assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=2e-6), (r, nq, key)
                    else:
                        assert a.shape == b.shape and np.array_equal(a, b), (r, nq, key, j)


def test_shard_partition_is_exact():
    ix, _, _ = _corpus()
    for world in (1, 2, 3, 8):
        b = ix.shard_bounds(world)
        assert b[0] == 0 and b[-1] == ix.n_docs and np.all(np.diff(b) >= 0)
        P = C = 0
        for r in range(world):
            s = ix.shard(r, world)
            P += int(s.post_doc.numel()); C += s.n_chunks
            assert s.doc_base == b[r] and s.n_docs == b[r + 1] - b[r]
            assert s.idf is ix.idf and s.avgdl == ix.avgdl            # global statistics are replicated
            if s.n_docs:
                assert int(s.post_doc.max()) < s.n_docs
        assert P == int(ix.post_doc.numel()) and C == ix.n_chunks
        # chunk balance within one document of the ideal cut
        sizes = [ix.shard(r, world).n_chunks for r in range(world)]
        assert max(sizes) - min(sizes) <= 2 * 64


def _run_odd(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from msretr.distributed import ShardedEngine
        se = ShardedEngine(None, 0, 0)
        # odd element counts: every segment must still be viewable as its dtype on the receiving side
        parts = [torch.arange(7, dtype=torch.int32) + rank, torch.arange(3, dtype=torch.float64) * (rank + 1),
                 torch.tensor([rank], dtype=torch.int32), torch.arange(5, dtype=torch.float32) - rank]
        got = se._allgather_bytes(parts)
        ret[rank] = [[t.clone().numpy() for t in g] for g in got]
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_packed_allgather_with_odd_sizes():
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_odd, args=(world, 31000 + os.getpid() % 2000, ret), nprocs=world, join=True)
    for r in range(world):
        for g in range(world):
            a = ret[r][g]
            assert a[0].tolist() == (np.arange(7) + g).tolist() and a[1].tolist() == (np.arange(3) * (g + 1.0)).tolist()
            assert a[2].tolist() == [g] and a[3].tolist() == (np.arange(5) - g).astype(np.float32).tolist()
