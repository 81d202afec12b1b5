"""Full-size checks (BASELINE.json shape: 1 M docs / 5 M x 768 f32 chunks / ~2.3e8 postings) on the MI355X.
The CPU oracle cannot finish these sizes in seconds, so the checks are (a) plain-PyTorch references of the
same arithmetic evaluated on the GPU (float64 for BM25 -> bitwise, float32 for the cosine -> 1e-5, the
north_star tolerance) and (b) size-independent properties: sortedness + tie rule, idempotence, scale
invariance of the cosine, max-pool monotonicity, sharded == unsharded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N_DOCS, N_CHUNKS = 1_000_000, 5_000_000


@pytest.fixture(scope="module")
def world():
    assert torch.cuda.is_available()
    from msretr.engine import DeviceEngine
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    dev = torch.device("cuda", 0)
    ix = synthetic_corpus(N_DOCS, n_chunks=N_CHUNKS, device=dev)
    terms, qvec = synthetic_queries(ix, 40, seed=99)
    eng = DeviceEngine(ix, max_queries=32, max_k=1000, rerank_max_docs=1000)
    yield dict(ix=ix, eng=eng, terms=terms, qvec=qvec, dev=dev, DeviceEngine=DeviceEngine)
    eng.close()


def _bm25_torch(ix, term_ids, k, min_score):
    """float64 PyTorch restatement of bm25_indexer.py:458-488 (one elementwise op per Python operation, so no
    fused multiply-add can sneak in)."""
    N = ix.n_docs
    acc = torch.zeros(N, dtype=torch.float64, device=ix.post_doc.device)
    touched = torch.zeros(N, dtype=torch.bool, device=acc.device)
    # a 0-dim DEVICE tensor: dividing by a Python scalar makes torch multiply by the reciprocal instead
    avgdl = torch.tensor(float(np.float32(ix.avgdl)), dtype=torch.float64, device=acc.device)
    cnt = {}
    for t in term_ids:
        cnt[t] = cnt.get(t, 0) + 1
    for t, f in cnt.items():
        if t < 0 or t >= ix.n_terms:
            continue
        lo, hi = int(ix.term_off[t]), int(ix.term_off[t + 1])
        if hi <= lo:
            continue
        d = ix.post_doc[lo:hi].long()
        tf = ix.post_tf[lo:hi].double()
        dl = ix.doc_len[d].double()
        idf = float(ix.idf[t])
        comp = (tf * (ix.k1 + 1)) / (tf + ix.k1 * ((1 - ix.b) + (ix.b * dl) / avgdl))
        acc[d] = acc[d] + (idf * comp) * float(f)
        touched[d] = True
    s = torch.where(touched & (acc >= min_score), acc, torch.full_like(acc, -float("inf")))
    key = torch.sort(-s, stable=True)                    # stable: ties keep ascending doc index
    n = int((s > -float("inf")).sum())
    return key.indices[:min(k, n)], -key.values[:min(k, n)]


def test_bm25_fullsize_bitwise_vs_torch_f64(world):
    ix, eng = world["ix"], world["eng"]
    tl = [ix.term_ids(t) for t in world["terms"][:12]]
    for k, ms in ((1000, 0.0), (100, -3.0)):
        doc, score, n = eng.bm25_topk(tl, k=k, min_score=ms)
        for i, t in enumerate(tl):
            ri, rs = _bm25_torch(ix, t, k, ms)
            assert int(n[i]) == len(ri)
            assert torch.equal(doc[i, :len(ri)].long(), ri)
            assert torch.equal(score[i, :len(ri)], rs)                   # float64, bitwise
            assert bool((torch.diff(score[i, :len(ri)]) <= 0).all())


def _dense_torch(ix, q, k):
    qn = q / torch.linalg.vector_norm(q)
    cos = torch.empty(ix.emb.shape[0], dtype=torch.float32, device=q.device)
    for s in range(0, ix.emb.shape[0], 1 << 20):
        e = ix.emb[s:s + (1 << 20)]
        cos[s:s + len(e)] = (e @ qn) / torch.linalg.vector_norm(e, dim=1)
    seg = torch.repeat_interleave(torch.arange(ix.n_docs, device=q.device), torch.diff(ix.doc_off.long()))
    best = torch.full((ix.n_docs,), -float("inf"), dtype=torch.float32, device=q.device)
    best = best.scatter_reduce(0, seg, cos, reduce="amax", include_self=True)
    top = torch.topk(best, k)
    return best, top.indices, top.values


def test_dense_fullsize_vs_torch_f32(world):
    ix, eng, qvec = world["ix"], world["eng"], world["qvec"]
    k = 100
    doc, score, chunk, n = eng.dense_topk(qvec[:32], k=k)
    assert bool((n == k).all())
    for i in (0, 7, 31):
        best, ti, tv = _dense_torch(ix, qvec[i], k)
        assert float((score[i] - best[doc[i].long()]).abs().max()) <= 1e-5       # every reported score is right
        assert float((score[i] - tv).abs().max()) <= 1e-5                          # and the list is the top-k
        missing = set(ti.tolist()) ^ set(doc[i].tolist())
        assert all(abs(float(best[d]) - float(tv[-1])) <= 2e-5 for d in missing)  # only boundary near-ties may swap
        lo = ix.doc_off[doc[i].long()].long(); hi = ix.doc_off[doc[i].long() + 1].long()
        assert bool(((chunk[i] >= lo) & (chunk[i] < hi)).all())
    # the planted source chunk of every query (synthetic_queries) is found, with a high cosine
    assert float(score[:, 0].min()) > 0.8


def test_stream_pass_fullsize_vs_torch_f32(world):
    """The kernel behind bench.py's `value` -- gemm_stream_kernel, ONE pass over the f32 rows for 128 queries, ~20 k row
    tiles at this size -- against the plain torch f32 reference at the full size: an exact hit, a
    near-duplicate, planted and random queries, at k = 100 and k = 10; a NaN query returns nothing and leaves every other
    64-query slice of the call bit for bit alone; the rerank of a 128-query ShardedEngine step equals that of the same
    queries served 64 at a time (the sweeps)."""
    from msretr.distributed import ShardedEngine
    from msretr.synthetic import synthetic_queries
    ix, dev = world["ix"], world["dev"]
    eng = world["DeviceEngine"](ix, max_queries=128, max_k=1000, rerank_max_docs=1000)
    assert eng.scan_width() == 128
    terms, qvec = synthetic_queries(ix, 128, seed=123)
    q = qvec.clone()
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    q[0] = ix.emb[4_321_987] * 3.0                                                 # exact hit: cosine 1
    noise = torch.nn.functional.normalize(torch.randn(768, generator=g), dim=0).to(dev)
    q[1] = (ix.emb[77] + 0.05 * noise) * 0.3                                       # near-duplicate of a row in the first tile
    q[2] = torch.randn(768, generator=g).to(dev) * 11.0                            # random direction: a flat score landscape
    q[127] = (ix.emb[N_CHUNKS - 1] + 0.2 * noise) * 6.0                            # near the LAST row of the matrix (last tile)
    picks = (0, 1, 2, 40, 64, 127)
    for k in (100, 10):
        doc, score, chunk, n = eng.dense_topk(q, k=k)
        assert bool((n == k).all())
        for i in picks:
            best, ti, tv = _dense_torch(ix, q[i], k)
            assert float((score[i] - best[doc[i].long()]).abs().max()) <= 1e-5    # every reported score is right
            assert float((score[i] - tv).abs().max()) <= 1e-5                      # and the list is the top-k
            missing = set(ti.tolist()) ^ set(doc[i].tolist())
            assert all(abs(float(best[d]) - float(tv[-1])) <= 2e-5 for d in missing)   # only boundary near-ties may swap
            lo = ix.doc_off[doc[i].long()].long(); hi = ix.doc_off[doc[i].long() + 1].long()
            assert bool(((chunk[i] >= lo) & (chunk[i] < hi)).all())
        assert abs(float(score[0, 0]) - 1.0) <= 1e-5 and int(chunk[0, 0]) == 4_321_987
        assert int(chunk[127, 0]) == N_CHUNKS - 1
    # the pass itself answered: its scores are exact f32 cosines, those of a 64-query call (K-split sweep, f16x2-split
    # products) differ from them in the last bits
    full = eng.dense_topk(q, k=100)
    sweep = eng.dense_topk(q[:64], k=100)
    assert not torch.equal(full[1][:64], sweep[1]) and float((full[1][:64] - sweep[1]).abs().max()) <= 1e-5
    # a NaN query: n == 0 for it; its own 64-query slice comes back from the gated sweeps (= a 64-query call, bit for bit),
    # the other slice keeps the pass' answer bit for bit
    qn = q.clone(); qn[100, 5] = float("nan")
    got = eng.dense_topk(qn, k=100)
    assert int(got[3][100]) == 0 and bool((got[3][torch.arange(128, device=dev) != 100] == 100).all())
    for a, b in zip(got, full):
        assert torch.equal(a[:64], b[:64])
    sw = eng.dense_topk(qn[64:], k=100)
    for a, b in zip(got, sw):
        assert torch.equal(a[64:], b)
    # one step of the whole path, 128 queries: the rerank equals that of the same queries served 64 at a time
    se = ShardedEngine(eng, 0, 0)
    tl = [ix.term_ids(t) for t in terms]
    one = se.search(tl, q, k1=1000, k2=100)
    halves = [se.search(tl[s:s + 64], q[s:s + 64], k1=1000, k2=100) for s in (0, 64)]
    for j in range(6):
        assert torch.equal(one["rerank"][j], torch.cat([h["rerank"][j] for h in halves]))
    for j in range(3):
        assert torch.equal(one["bm25"][j], torch.cat([h["bm25"][j] for h in halves]))
    hd = [torch.cat([h["dense"][j] for h in halves]) for j in range(4)]
    assert torch.equal(one["dense"][3], hd[3]) and float((one["dense"][1] - hd[1]).abs().max()) <= 1e-5
    assert float((one["dense"][0] == hd[0]).float().mean()) > 0.99
    eng.close()


def test_stream256_pass_fullsize_vs_torch_f32(world):
    """The kernel behind bench.py's `value` since round 3 -- gemm_stream256_kernel on the fragment-order copy of the rows, 256
    queries per workgroup pass -- against the plain torch f32 reference at the full size: an exact hit, a near-duplicate,
    random and planted queries, the last row of the matrix; then 1024 queries in ONE call (four groups of 256 sharing the
    rows of a launch): the first group bit for bit what the 256-query call returned, queries of the other groups against
    torch, a NaN query that returns nothing and leaves every other 64-query slice of the call bit for bit alone."""
    from msretr.synthetic import synthetic_queries
    ix, dev = world["ix"], world["dev"]
    eng = world["DeviceEngine"](ix, max_queries=1024, max_k=100, rerank_max_docs=0)
    assert eng.scan_width() == 256
    _, qvec = synthetic_queries(ix, 1024, seed=321)
    q = qvec.clone()
    g = torch.Generator(device="cpu"); g.manual_seed(6)
    noise = torch.nn.functional.normalize(torch.randn(768, generator=g), dim=0).to(dev)
    q[0] = ix.emb[3_210_987] * 2.5                                                 # exact hit: cosine 1
    q[1] = (ix.emb[91] + 0.05 * noise) * 0.4                                       # near-duplicate of a row in the first tile
    q[2] = torch.randn(768, generator=g).to(dev) * 9.0                             # random direction
    q[255] = (ix.emb[N_CHUNKS - 1] + 0.2 * noise) * 6.0                            # near the LAST row of the matrix
    q[700] = ix.emb[1_234_567] * 0.7                                               # an exact hit in the third group
    q[1023] = (ix.emb[N_CHUNKS - 2] + 0.1 * noise) * 2.0                           # the last tile again, last query of the call

    def check(i, doc, score, chunk, k):
        best, ti, tv = _dense_torch(ix, q[i], k)
        assert float((score - best[doc.long()]).abs().max()) <= 1e-5              # every reported score is right
        assert float((score - tv).abs().max()) <= 1e-5                            # and the list is the top-k
        missing = set(ti.tolist()) ^ set(doc.tolist())
        assert all(abs(float(best[d]) - float(tv[-1])) <= 2e-5 for d in missing)   # only boundary near-ties may swap
        lo = ix.doc_off[doc.long()].long(); hi = ix.doc_off[doc.long() + 1].long()
        assert bool(((chunk >= lo) & (chunk < hi)).all())

    for k in (100, 10):
        doc, score, chunk, n = eng.dense_topk(q[:256], k=k)
        assert eng.dense_path() == 256 and bool((n == k).all())
        for i in (0, 1, 2, 130, 255):
            check(i, doc[i], score[i], chunk[i], k)
        assert abs(float(score[0, 0]) - 1.0) <= 1e-5 and int(chunk[0, 0]) == 3_210_987
        assert int(chunk[255, 0]) == N_CHUNKS - 1
    one = eng.dense_topk(q[:256], k=100)
    four = eng.dense_topk(q, k=100)
    assert eng.dense_path() == 256 and bool((four[3] == 100).all())
    for a, b in zip(four, one):
        assert torch.equal(a[:256], b)                                             # exact f32 finish: the same bits
    for i in (300, 511, 700, 1023):
        check(i, four[0][i], four[1][i], four[2][i], 100)
    assert abs(float(four[1][700, 0]) - 1.0) <= 1e-5 and int(four[2][700, 0]) == 1_234_567
    qn = q.clone(); qn[600, 9] = float("nan")
    got = eng.dense_topk(qn, k=100)
    keep = torch.ones(1024, dtype=torch.bool, device=dev); keep[576:640] = False    # the NaN query's 64-query slice
    assert int(got[3][600]) == 0 and bool((got[3][torch.arange(1024, device=dev) != 600] == 100).all())
    for a, b in zip(got, four):
        assert torch.equal(a[keep], b[keep])
    eng.close()


def test_dense_fullsize_f16split_vs_exact(world):
    """The default (f16-split) scan against the exact f32 MFMA scan on the full corpus: same top-100 up to
    near-ties, scores within the proven bound."""
    ix, eng, qvec = world["ix"], world["eng"], world["qvec"]
    assert eng.scan_arith() == "f16x2"
    exact = world["DeviceEngine"](ix, max_queries=32, max_k=100, rerank_max_docs=0, scan_variant=2)
    assert exact.scan_arith() == "f32"
    a = eng.dense_topk(qvec[:32], k=100)
    b = exact.dense_topk(qvec[:32], k=100)
    err = float((a[1] - b[1]).abs().max())
    print(f"max |f16split - exact f32| over 32 x 100 scores: {err:.3e}")
    assert err <= 8e-6
    assert bool(((a[0] == b[0]) | ((a[1] - b[1]).abs() <= 8e-6)).all())
    exact.close()


def test_dense_fullsize_properties(world):
    eng, qvec = world["eng"], world["qvec"]
    a = eng.dense_topk(qvec[:32], k=100)
    b = eng.dense_topk(qvec[:32], k=100)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                                                   # idempotent, bit for bit
    assert bool((torch.diff(a[1], dim=1) <= 0).all())                              # sorted
    eq = torch.diff(a[1], dim=1) == 0
    assert bool((torch.diff(a[0], dim=1)[eq] > 0).all())                           # ties: ascending index
    c = eng.dense_topk(qvec[:32] * 7.5, k=100)                                     # cosine is scale invariant
    assert float((c[1] - a[1]).abs().max()) <= 2e-6
    assert float((c[0] == a[0]).float().mean()) > 0.99
    d1 = eng.dense_topk(qvec[:8], k=100, max_chunks_per_doc=1)                     # max over fewer chunks is smaller
    d3 = eng.dense_topk(qvec[:8], k=100, max_chunks_per_doc=3)
    # (8-query and 32-query launches sum in a different order: allow the rounding of two f32 sums)
    assert bool((d1[1] <= d3[1] + 2e-6).all()) and bool((d3[1] <= a[1][:8] + 2e-6).all())
    # a batch of 40 (one 64-wide sweep of the K-split kernel) against the same queries issued as 32 + 8 (narrow kernel):
    # another summation order, so equal up to f32 rounding and near-tie swaps
    e = eng.dense_topk(world["qvec"][:40], k=100)
    f = eng.dense_topk(world["qvec"][32:40], k=100)
    ref_doc, ref_score = torch.cat([a[0], f[0]]), torch.cat([a[1], f[1]])
    assert float((e[1] - ref_score).abs().max()) <= 2e-6
    assert bool(((e[0] == ref_doc) | ((e[1] - ref_score).abs() <= 2e-6)).all())
    # and the wide sweep is itself idempotent, bit for bit
    e2 = eng.dense_topk(world["qvec"][:40], k=100)
    assert torch.equal(e[0], e2[0]) and torch.equal(e[1], e2[1])


def test_fullsize_two_shards_equal_unsharded(world):
    ix, eng, qvec = world["ix"], world["eng"], world["qvec"]
    tl = [ix.term_ids(t) for t in world["terms"][:8]]
    fb = eng.bm25_topk(tl, k=1000)
    fd = eng.dense_topk(qvec[:8], k=100)
    fr = eng.rerank(qvec[:8], fb[0], fb[1], fb[2])
    shards = [ix.shard(r, 2) for r in range(2)]
    engs = [world["DeviceEngine"](s, max_queries=8, max_k=1000, rerank_max_docs=1000) for s in shards]
    glob = lambda t, base: torch.where(t >= 0, t + base, t)
    pb = [e.bm25_topk(tl, k=1000) for e in engs]
    pd_ = [e.dense_topk(qvec[:8], k=100) for e in engs]
    mb = engs[0].merge_topk(torch.stack([glob(p[0], s.doc_base) for p, s in zip(pb, shards)]),
                            torch.stack([p[1] for p in pb]), torch.stack([p[2] for p in pb]), 1000)
    md = engs[0].merge_topk(torch.stack([glob(p[0], s.doc_base) for p, s in zip(pd_, shards)]),
                            torch.stack([p[1] for p in pd_]), torch.stack([p[3] for p in pd_]), 100)
    for x, y in zip(mb, fb):
        assert torch.equal(x, y)
    for x, y in zip(md, (fd[0], fd[1], fd[3])):
        assert torch.equal(x, y)
    parts = [e.rerank_gather(qvec[:8], mb[0], mb[2], doc_base=s.doc_base, row_base=s.row_base) for e, s in zip(engs, shards)]
    cos = parts[0][0].view(torch.int32) | parts[1][0].view(torch.int32)
    meta = parts[0][1] | parts[1][1]
    mr = engs[1].rerank_fuse(mb[0], mb[1], mb[2], cos.view(torch.float32), meta)
    for x, y in zip(mr, fr):
        assert torch.equal(x, y)
    for e in engs:
        e.close()


def test_fullsize_batched_bf16_equals_exact(world):
    """bf16 candidate sweep + f32 rescoring returns the exact path's top-100 at 1 M docs / 5 M chunks."""
    eng, qvec = world["eng"], world["qvec"]
    eng.enable_bf16()
    ex = eng.dense_topk(qvec[:40], k=100)
    got = eng.dense_topk_batched(qvec[:40], k=100)
    assert torch.equal(got[3], ex[3])
    assert float((got[1] - ex[1]).abs().max()) <= 2e-6
    assert bool(((got[0] == ex[0]) | ((got[1] - ex[1]).abs() <= 2e-6)).all())
    assert float((got[0] == ex[0]).float().mean()) > 0.995


def test_fullsize_gemm_1024_queries_equals_exact(world):
    """BASELINE configs[4] shape on one GPU: 1024 queries x 5 M chunks through the tiled GEMM path; a sample of the
    batch is compared with the exact f32 scan, the rest through size-independent properties."""
    eng, ix, dev = world["eng"], world["ix"], world["dev"]
    eng.enable_bf16()
    assert eng.batch_gemm_ok()
    g = torch.Generator(device="cpu"); g.manual_seed(21)
    rows = torch.randint(0, N_CHUNKS, (1024,), generator=g)
    noise = torch.nn.functional.normalize(torch.randn((1024, 768), generator=g), dim=1).to(dev)
    q = (ix.emb[rows.to(dev)] + 0.5 * noise) * 9.0
    q[0] = ix.emb[2_468_013] * 4.0                                                 # exact hit, first query group
    q[600] = torch.randn(768, generator=g).to(dev) * 3.0                           # random direction: a flat score landscape
    q[1023] = (ix.emb[N_CHUNKS - 1] + 0.1 * noise[0]) * 0.6                        # the last row of the last tile, last group
    got = eng.dense_topk_batched(q, k=100)
    # the comparator above 128 queries is NOT another HIP path: plain torch f32 (reranker_api.py:285 arithmetic, per-document
    # max :370) for queries of the first / a middle / the last group of 256 -- an exact hit, a flat landscape, the last tile
    for i in (0, 1, 255, 256, 600, 777, 1022, 1023):
        best, ti, tv = _dense_torch(ix, q[i], 100)
        assert float((got[1][i] - best[got[0][i].long()]).abs().max()) <= 1e-5    # every reported score is right
        assert float((got[1][i] - tv).abs().max()) <= 1e-5                        # and the list is the top-k
        missing = set(ti.tolist()) ^ set(got[0][i].tolist())
        assert all(abs(float(best[d]) - float(tv[-1])) <= 2e-5 for d in missing)   # only boundary near-ties may swap
    assert abs(float(got[1][0, 0]) - 1.0) <= 1e-5 and int(got[2][0, 0]) == 2_468_013
    assert int(got[2][1023, 0]) == N_CHUNKS - 1
    assert bool((got[3] == 100).all())
    assert bool((torch.diff(got[1], dim=1) <= 0).all())                            # sorted
    eq = torch.diff(got[1], dim=1) == 0
    assert bool((torch.diff(got[0], dim=1)[eq] > 0).all())                         # ties: ascending index
    planted = torch.ones(1024, dtype=torch.bool, device=dev); planted[600] = False
    assert float(got[1][planted, 0].min()) > 0.8                                   # the planted chunk is found
    lo = ix.doc_off[got[0].long()].long(); hi = ix.doc_off[got[0].long() + 1].long()
    assert bool(((got[2] >= lo) & (got[2] < hi)).all())                            # arg-max chunk inside its document
    sel = torch.arange(0, 1024, 8, device=dev)                                     # 128 of the queries against the exact scan
    ex = eng.dense_topk(q[sel], k=100)
    assert float((got[1][sel] - ex[1]).abs().max()) <= 2e-6
    assert bool(((got[0][sel] == ex[0]) | ((got[1][sel] - ex[1]).abs() <= 2e-6)).all())
    assert float((got[0][sel] == ex[0]).float().mean()) > 0.995
    again = eng.dense_topk_batched(q, k=100)                                       # idempotent, bit for bit
    for x, y in zip(got, again):
        assert torch.equal(x, y)


def _rerank_expected(ix, q, cand_doc, cand_bm25, n):
    """The /rerank arithmetic (reranker_api.py:27-63 fetch of <= 10 chunks per candidate, :285 f32 cosine as sklearn computes it,
    :360-372 chain) for ONE query at the full size: rows gathered and multiplied by plain torch f32 on the GPU, the float64
    chain by the oracle's Python-float restatement (oracle.rerank_ref.chain_from_cosines, pinned to rerank_chain.json).
    The synthetic corpus has no urlsDB strings: every document is its own URL group and none is missing."""
    from oracle import rerank_ref
    docs = cand_doc[:n].long()
    order = torch.argsort(docs)                                        # the reference's rows come out in doc_id order
    docs, bm = docs[order], cand_bm25[:n][order]
    lo = ix.doc_off[docs].long(); cnt = torch.clamp(ix.doc_off[docs + 1].long() - lo, max=10)
    keep = cnt > 0
    docs, bm, lo, cnt = docs[keep], bm[keep], lo[keep], cnt[keep]
    rows = torch.cat([torch.arange(int(a), int(a) + int(c), device=q.device) for a, c in zip(lo.tolist(), cnt.tolist())])
    e = ix.emb[rows]
    en = torch.linalg.vector_norm(e, dim=1); en[en == 0] = 1.0
    qn = torch.linalg.vector_norm(q); qn = qn if float(qn) != 0.0 else torch.ones_like(qn)
    cos = ((e / en[:, None]) @ (q / qn)).float().cpu().numpy()
    pooled, _ = rerank_ref.chain_from_cosines(docs.tolist(), cnt.tolist(), bm.tolist(), cos)
    first = dict(zip(docs.tolist(), lo.tolist()))
    pooled = sorted(pooled, key=lambda x: (-x[1], x[0]))
    return [(d, s, o, first[d] + b) for d, s, o, b in pooled], int(cnt.sum())


def _check_rerank(ix, q, b, r, i, tol=5e-6):
    """row i of a rerank result r = (doc, score, orig, chunk, n, rows) against _rerank_expected."""
    exp, n_rows = _rerank_expected(ix, q, b[0][i], b[1][i], int(b[2][i]))
    n = int(r[4][i])
    assert n == len(exp) and int(r[5][i]) == n_rows
    gd, gs, go, gc = r[0][i, :n].tolist(), r[1][i, :n].tolist(), r[2][i, :n].tolist(), r[3][i, :n].tolist()
    assert float(np.abs(np.array(gs) - np.array([e[1] for e in exp])).max()) <= tol
    assert sorted(gd) == sorted(e[0] for e in exp)
    emap = {e[0]: e for e in exp}
    swaps = 0
    for j, d in enumerate(gd):
        e = emap[d]
        assert abs(gs[j] - e[1]) <= tol and abs(go[j] - e[2]) <= 1e-12
        if gc[j] != e[3]:                                             # arg-max chunk: may differ only on a near-tie inside the doc
            swaps += 1
        if d != exp[j][0]:                                            # rank swap: only between scores within rounding
            assert abs(exp[j][1] - e[1]) <= 2 * tol
    assert swaps <= max(1, n // 100)
    assert all(gs[j] >= gs[j + 1] for j in range(n - 1))


def test_fullsize_single_query(world):
    """BASELINE configs[2] as worded -- ONE query on 1 M docs / 5 M chunks: the Q = 1 instantiation of the K-split sweep (the
    kernel behind p50_latency_ms_single_query) and the whole hybrid step with one query, against torch f64 (BM25, bitwise),
    torch f32 (dense, 1e-5) and the oracle's chain (rerank, 5e-6)."""
    from msretr.distributed import ShardedEngine
    ix, eng, dev = world["ix"], world["eng"], world["dev"]
    g = torch.Generator(device="cpu"); g.manual_seed(77)
    noise = torch.nn.functional.normalize(torch.randn(768, generator=g), dim=0).to(dev)
    singles = [world["qvec"][3], ix.emb[1_357_911] * 2.0, (ix.emb[N_CHUNKS - 1] + 0.1 * noise) * 5.0,
               torch.randn(768, generator=g).to(dev) * 7.0]
    for j, q1 in enumerate(singles):
        doc, score, chunk, n = eng.dense_topk(q1.reshape(1, 768), k=100)
        assert int(n[0]) == 100
        best, ti, tv = _dense_torch(ix, q1, 100)
        assert float((score[0] - best[doc[0].long()]).abs().max()) <= 1e-5
        assert float((score[0] - tv).abs().max()) <= 1e-5
        missing = set(ti.tolist()) ^ set(doc[0].tolist())
        assert all(abs(float(best[d]) - float(tv[-1])) <= 2e-5 for d in missing)
        if j == 1:
            assert abs(float(score[0, 0]) - 1.0) <= 1e-5 and int(chunk[0, 0]) == 1_357_911
        if j == 2:
            assert int(chunk[0, 0]) == N_CHUNKS - 1
    se = ShardedEngine(eng, 0, 0)
    for i in (0, 5, 11):
        tl = [ix.term_ids(world["terms"][i])]
        q1 = world["qvec"][i:i + 1].contiguous()
        out = se.search(tl, q1, k1=1000, k2=100)
        ri, rs = _bm25_torch(ix, tl[0], 1000, 0.0)
        b = out["bm25"]
        assert int(b[2][0]) == len(ri) and torch.equal(b[0][0, :len(ri)].long(), ri) and torch.equal(b[1][0, :len(ri)], rs)
        best, ti, tv = _dense_torch(ix, q1[0], 100)
        d = out["dense"]
        assert float((d[1][0] - tv).abs().max()) <= 1e-5 and float((d[1][0] - best[d[0][0].long()]).abs().max()) <= 1e-5
        _check_rerank(ix, q1[0], b, out["rerank"], 0)


def test_fullsize_rerank_vs_oracle_chain(world):
    """Rerank / fuse at the benchmark's size and batch shape (256 queries per step on the headline path): rows of the step's
    rerank output against the reference's arithmetic (reranker_api.py:285, 357-372) restated by torch f32 + the oracle."""
    from msretr.distributed import ShardedEngine
    from msretr.synthetic import synthetic_queries
    ix = world["ix"]
    eng = world["DeviceEngine"](ix, max_queries=256, max_k=1000, rerank_max_docs=1000)
    terms, qvec = synthetic_queries(ix, 256, seed=4242)
    se = ShardedEngine(eng, 0, 0)
    tl = [ix.term_ids(t) for t in terms]
    out = se.search(tl, qvec, k1=1000, k2=100)
    assert eng.dense_path() == 256
    for i in (0, 63, 128, 200, 255):
        _check_rerank(ix, qvec[i], out["bm25"], out["rerank"], i)
    # rerank_keep truncates the lists AND the counts
    cut = se.search(tl, qvec, k1=1000, k2=100, rerank_keep=100)["rerank"]
    assert tuple(cut[0].shape) == (256, 100) and int(cut[4].max()) <= 100
    assert torch.equal(cut[0], out["rerank"][0][:, :100]) and torch.equal(cut[1], out["rerank"][1][:, :100])
    eng.close()
