"""HIP path vs the CPU oracle and vs the reference-generated goldens.  Every call goes through the C ABI
(libmsretr.so).  Run on the GPU box: pytest -m gpu."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(G, name), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def mods():
    import msretr
    from msretr.engine import DeviceEngine
    from msretr.index import CorpusIndex
    from oracle import bm25_ref, dense_ref, rerank_ref
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return dict(DeviceEngine=DeviceEngine, CorpusIndex=CorpusIndex, bm25_ref=bm25_ref, dense_ref=dense_ref,
                rerank_ref=rerank_ref, msretr=msretr)


def _ix_from_npz(CorpusIndex, z):
    return CorpusIndex(doc_ids=z["doc_ids"], doc_len=z["doc_len"], term_off=z["term_off"], post_doc=z["post_doc"],
                       post_tf=z["post_tf"], idf=z["idf"], avgdl=float(z["avgdl"]), total_docs=int(z["total_docs"]))


# ------------------------------------------------------------------------------------------------ BM25
@pytest.mark.parametrize("name", ["bm25_random_a", "bm25_random_b"])
def test_bm25_golden_random(mods, name):
    z = dict(np.load(os.path.join(G, name + ".npz")))
    meta = _load(name + ".json")
    ix = _ix_from_npz(mods["CorpusIndex"], z)
    eng = mods["DeviceEngine"](ix, max_queries=8, max_k=1000)
    missing = set(meta["missing_from_urlsdb"])
    for min_score in sorted({q["min_score"] for q in meta["queries"]}):
        for top_k in sorted({q["top_k"] for q in meta["queries"]}):
            qs = [q for q in meta["queries"] if q["top_k"] == top_k and q["min_score"] == min_score]
            if not qs:
                continue
            doc, score, n = eng.bm25_topk([q["terms"] for q in qs], k=top_k, min_score=min_score)
            doc, score, n = doc.cpu().numpy(), score.cpu().numpy(), n.cpu().numpy()
            for i, q in enumerate(qs):
                ids = [int(z["doc_ids"][d]) for d in doc[i, :n[i]]]
                sc = [float(s) for s in score[i, :n[i]]]
                # engine level == oracle before the urlsDB join (bitwise float64)
                oi, os_ = mods["bm25_ref"].topk(z, q["terms"], top_k, min_score)
                assert doc[i, :n[i]].tolist() == oi.tolist()
                assert sc == os_.tolist()
                assert np.all(doc[i, n[i]:] == -1) and np.all(np.isneginf(score[i, n[i]:]))
                # reference golden == engine result after dropping documents absent from urlsDB
                keep = [(d, s) for d, s in zip(ids, sc) if d not in missing]
                assert [d for d, _ in keep] == q["doc_id"]
                assert [s for _, s in keep] == q["score"]
    eng.close()


def test_bm25_known_answers(mods):
    from msretr.bm25 import BM25
    for case in _load("bm25_kat.json"):
        postings = {t: [tuple(p) for p in pl] for t, pl in case["postings"].items()}
        doc_len = {int(d): l for d, l in case["doc_len"].items()}
        urls_db = {int(d): ("http://x/%s" % d, v[0], v[1]) for d, v in case["urls_db"].items()}
        ix = mods["CorpusIndex"].from_tables(postings, doc_len, case["idf_f32"], case["avgdl_f32"],
                                              total_docs=case["total_docs"], urls_db=urls_db,
                                              k1=case["k1"], b=case["b"])
        bm = BM25(ix, k1=case["k1"], b=case["b"], tokenizer=lambda s: s.split(), max_queries=4, max_k=16)
        for q in case["queries"]:
            got = bm.search(" ".join(q["terms"]), top_k=min(q["top_k"], 16), min_score=q["min_score"])
            assert got == q["expected"], (case["name"], q["terms"])
        bm.engine.close()


def test_bm25_synthetic_vs_oracle(mods):
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    ix = synthetic_corpus(30000, n_chunks=0, n_terms=20000, seed=5)
    terms, _ = synthetic_queries(ix, 40, seed=6)
    terms[3] = [7, 7, 7, 0, 19999, 250000, -1]          # repeats, rare, out-of-range and negative ids
    terms[4] = []                                        # empty query
    terms[5] = [0]                                       # only the negative-idf term
    terms[6] = [int(t) for t in np.random.default_rng(1).choice(5000, size=41, replace=False)] + [3, 3]   # many terms
    z = {k: (getattr(ix, k).cpu().numpy() if torch.is_tensor(getattr(ix, k)) else getattr(ix, k))
         for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
    z["avgdl"] = ix.avgdl
    eng = mods["DeviceEngine"](ix, max_queries=16, max_k=1000)      # 40 queries -> 3 internal slices
    for k, ms in ((1000, 0.0), (100, 0.0), (7, -50.0), (1000, 2.5)):
        doc, score, n = eng.bm25_topk(terms, k=k, min_score=ms)
        doc, score, n = doc.cpu().numpy(), score.cpu().numpy(), n.cpu().numpy()
        for i, t in enumerate(terms):
            oi, os_ = mods["bm25_ref"].topk(z, t, k, ms)
            assert n[i] == len(oi), (i, k, ms)
            assert doc[i, :n[i]].tolist() == oi.tolist(), (i, k, ms)
            assert score[i, :n[i]].tolist() == os_.tolist(), (i, k, ms)
    eng.close()


def test_bm25_many_tiles_common_rare_and_tied_terms(mods):
    """80 k and 300 k documents (79 / 293 document tiles): queries with the most common term (long slices: the streaming
    loop behind the prefetch), the rarest terms only (lists of <= 64 postings are read whole and masked; a few dozen
    candidates in all), a repeated term, an empty query, min_score above and below most scores, k = 10 and 1000; then
    70 000 postings that all tie.  Bit for bit against the oracle."""
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    for n_docs, seed in ((80_000, 21), (300_000, 22)):
        ix = synthetic_corpus(n_docs, n_chunks=0, n_terms=50_000, seed=seed)
        z = {k: (getattr(ix, k).cpu().numpy() if torch.is_tensor(getattr(ix, k)) else getattr(ix, k))
             for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
        z["avgdl"] = ix.avgdl
        df = np.diff(z["term_off"])
        rare = [int(t) for t in np.argsort(np.where(df > 0, df, 1 << 40), kind="stable")[:6]]   # the six shortest lists
        common = int(np.argmax(df))
        terms, _ = synthetic_queries(ix, 12, seed=seed + 1)
        terms[0] = [common]
        terms[1] = rare[:3]                                  # a few dozen candidates in all
        terms[2] = [common, common, rare[3]]
        terms[3] = []
        eng = mods["DeviceEngine"](ix, max_queries=8, max_k=1000)       # 12 queries -> 2 internal slices
        for k, ms in ((1000, 0.0), (10, 0.0), (1000, 3.0), (100, -5.0)):
            doc, score, n = [x.cpu().numpy() for x in eng.bm25_topk(terms, k=k, min_score=ms)]
            for i, t in enumerate(terms):
                oi, os_ = mods["bm25_ref"].topk(z, t, k, ms)
                assert n[i] == len(oi), (n_docs, i, k, ms)
                assert doc[i, :n[i]].tolist() == oi.tolist() and score[i, :n[i]].tolist() == os_.tolist(), (n_docs, i, k, ms)
        eng.close()
    # every posting of the only term ties
    N = 70_000
    ix = mods["CorpusIndex"](doc_ids=np.arange(N, dtype=np.int64), doc_len=np.full(N, 7, np.int32),
                             term_off=np.array([0, N], np.int64), post_doc=np.arange(N, dtype=np.int32),
                             post_tf=np.full(N, 2, np.int32), idf=np.array([0.8], np.float32), avgdl=7.0, total_docs=N)
    eng = mods["DeviceEngine"](ix, max_queries=2, max_k=1000)
    doc, score, n = [x.cpu().numpy() for x in eng.bm25_topk([[0]], k=1000)]
    assert n[0] == 1000 and doc[0].tolist() == list(range(1000)) and len(set(score[0].tolist())) == 1
    eng.close()


def test_bm25_select_window_pass_and_its_fallbacks(mods):
    """The BM25 select's first pass histograms a window of 16 octaves below a bound of the query's scores.  Scores far below it
    (a term whose idf is 1e-7 of another's: the k-th score falls into the clamped lowest bin), scores spread over 30 octaves,
    a bound that is loose by orders of magnitude (high query term frequencies on terms most documents lack), and a negative
    min_score (the general two-pass path) -- all bit for bit against the oracle, k below, at and above the strong documents."""
    rng = np.random.default_rng(31)
    N = 6000
    doc_len = rng.integers(5, 400, size=N).astype(np.int32)
    lists = [np.sort(rng.choice(N, size=12, replace=False)),           # term 0: strong, rare
             np.sort(rng.choice(N, size=4500, replace=False)),         # term 1: idf 1e-7 of term 0's
             np.sort(rng.choice(N, size=3000, replace=False)),         # term 2: in between
             np.arange(N)]                                             # term 3: every document, tiny idf
    idf = np.array([6.0, 6e-7, 2e-3, 3e-9], np.float32)
    term_off = np.zeros(len(lists) + 1, np.int64); term_off[1:] = np.cumsum([len(l) for l in lists])
    post_doc = np.concatenate(lists).astype(np.int32)
    post_tf = rng.integers(1, 9, size=len(post_doc)).astype(np.int32)
    ix = mods["CorpusIndex"](doc_ids=np.arange(N, dtype=np.int64), doc_len=doc_len, term_off=term_off, post_doc=post_doc,
                             post_tf=post_tf, idf=idf, avgdl=float(doc_len.mean()), total_docs=N)
    z = {k: (getattr(ix, k).cpu().numpy() if torch.is_tensor(getattr(ix, k)) else getattr(ix, k))
         for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
    z["avgdl"] = ix.avgdl
    terms = [[0, 1], [0, 1, 2, 3], [1], [3], [0], [0] * 40 + [3], [2, 3, 3, 3], []]
    eng = mods["DeviceEngine"](ix, max_queries=8, max_k=1000)
    for k, ms in ((5, 0.0), (12, 0.0), (100, 0.0), (1000, 0.0), (1000, 1e-8), (100, -1.0)):
        doc, score, n = [x.cpu().numpy() for x in eng.bm25_topk(terms, k=k, min_score=ms)]
        for i, t in enumerate(terms):
            oi, os_ = mods["bm25_ref"].topk(z, t, k, ms)
            assert n[i] == len(oi), (i, k, ms)
            assert doc[i, :n[i]].tolist() == oi.tolist() and score[i, :n[i]].tolist() == os_.tolist(), (i, k, ms)
    eng.close()


def test_bm25_negative_idf_terms_are_looked_up_in_query_order(mods):
    """The scoring kernel does not stream the long lists with negative idf (min_score >= 0): their contributions are looked up
    in dense tables for the documents the other terms touch, and must enter every sum at the term's position in the query.
    70 such terms (only 64 get a table: the others are streamed), one dense term with idf exactly 0 (not prunable: its documents
    score 0.0 >= 0.0), a dense positive term, a SHORT list with a negative idf (no table), medium and short positive lists;
    queries put the negative terms first, last, in between, repeated (query frequency 2), alone, and beyond the four
    prefetched streamed terms; min_score on both sides of 0.  Bit for bit against the oracle."""
    rng = np.random.default_rng(2024)
    N, lists, idf = 5000, [], []
    def add(p_in, value):
        docs = np.nonzero(rng.random(N) < p_in)[0].astype(np.int32)
        lists.append(docs); idf.append(value)
    for _ in range(70):
        add(0.6, -rng.uniform(0.05, 0.9))                    # 0..69: ~3000 postings each, idf < 0
    add(0.7, 0.0)                                            # 70: dense, idf exactly 0
    add(0.55, 0.3)                                           # 71: dense, idf > 0
    for _ in range(8):
        add(0.05, rng.uniform(0.5, 3.0))                     # 72..79: ~250 postings (probed ranges)
    add(0.006, -0.4)                                         # 80: ~30 postings with a NEGATIVE idf: streamed, no table
    for _ in range(9):
        add(0.006, rng.uniform(1.0, 4.0))                    # 81..89: ~30 postings
    term_off = np.zeros(len(lists) + 1, np.int64); term_off[1:] = np.cumsum([len(x) for x in lists])
    post_doc = np.concatenate(lists)
    post_tf = rng.integers(1, 6, size=len(post_doc)).astype(np.int32)
    doc_len = rng.integers(5, 900, size=N).astype(np.int32)
    avgdl = float(np.float32(doc_len.mean()))
    z = dict(doc_ids=np.arange(N, dtype=np.int64) * 2 + 1, doc_len=doc_len, term_off=term_off, post_doc=post_doc, post_tf=post_tf,
             idf=np.asarray(idf, np.float32), avgdl=avgdl)
    ix = mods["CorpusIndex"](total_docs=N, **z)
    eng = mods["DeviceEngine"](ix, max_queries=16, max_k=1000)
    queries = [
        [3, 72, 81],                       # negative first
        [72, 81, 3],                       # ... last (what preprocess_query's appended city term looks like)
        [72, 3, 81, 5, 82],                # ... in between, two of them
        [3, 3, 72],                        # query frequency 2 on a looked-up term
        [3], [3, 5, 7],                    # negative terms only: no candidate at min_score >= 0
        [69, 68, 67, 66, 65, 64, 72],      # (some of) the negative terms without a table: streamed
        [70, 3], [3, 70, 72],              # idf == 0: every document of term 70 is a candidate
        [71, 3, 72], [80, 81], [80, 3], [80],
        [72, 1, 73, 2, 74, 3, 75, 4, 76, 5, 77, 6, 81, 7],          # lookups beyond the four prefetched streamed terms
        [int(t) for t in rng.permutation(90)[:50]],               # 50 terms of every kind
        [int(t) for t in rng.permutation(90)[:64]],               # MSR_MAX_QUERY_TERMS
        [81, 82, 83],                      # no negative term at all
        [200, -3, 3, 72],                  # unknown ids around a looked-up term
    ]
    for k, ms in ((1000, 0.0), (10, 0.0), (1000, 0.75), (1000, -0.5), (100, -100.0)):
        for s0 in range(0, len(queries), 16):
            qs = queries[s0:s0 + 16]
            doc, score, n = [x.cpu().numpy() for x in eng.bm25_topk(qs, k=k, min_score=ms)]
            for i, t in enumerate(qs):
                oi, os_ = mods["bm25_ref"].topk(z, t, k, ms)
                assert n[i] == len(oi), (s0 + i, k, ms)
                assert doc[i, :n[i]].tolist() == oi.tolist() and score[i, :n[i]].tolist() == os_.tolist(), (s0 + i, k, ms)
    eng.close()


def test_bm25_massive_ties(mods):
    """All documents identical => every score equal: the k lowest doc indices must come back in order."""
    N = 20000
    ix = mods["CorpusIndex"](doc_ids=np.arange(N, dtype=np.int64) * 3 + 5, doc_len=np.full(N, 10, np.int32),
                             term_off=np.array([0, N], np.int64), post_doc=np.arange(N, dtype=np.int32),
                             post_tf=np.ones(N, np.int32), idf=np.array([1.5], np.float32), avgdl=10.0, total_docs=N)
    eng = mods["DeviceEngine"](ix, max_queries=2, max_k=1000)
    doc, score, n = eng.bm25_topk([[0], [0, 0]], k=1000)
    doc, score, n = doc.cpu().numpy(), score.cpu().numpy(), n.cpu().numpy()
    assert n.tolist() == [1000, 1000]
    assert doc[0].tolist() == list(range(1000)) and doc[1].tolist() == list(range(1000))
    assert len(set(score[0].tolist())) == 1
    eng.close()


# ------------------------------------------------------------------------------------------------ dense
def _rand_chunked(rng, n_docs, max_ch, big=()):
    n = rng.integers(0, max_ch + 1, size=n_docs)
    for pos, size in big:
        n[pos] = size
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb = rng.standard_normal((C, 768)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    return doc_off, emb


def _check_dense(mods, eng, doc_off, emb, q, k, mc, got):
    doc, score, chunk, n = [x.cpu().numpy() for x in got]
    for i in range(q.shape[0]):
        best, arg = mods["dense_ref"].doc_scores(emb, doc_off, q[i], mc)
        oi, os_, oa = mods["dense_ref"].quick_search(emb, doc_off, q[i], k, mc)
        assert n[i] == len(oi)
        np.testing.assert_allclose(score[i, :n[i]], os_, rtol=0, atol=1e-5)        # north_star: 1e-5 fp32
        np.testing.assert_allclose(score[i, :n[i]], best[doc[i, :n[i]]], rtol=0, atol=1e-5)
        assert np.all(np.diff(score[i, :n[i]]) <= 0)
        # same documents, except for swaps among scores closer than the tolerance
        diff = set(doc[i, :n[i]].tolist()) ^ set(oi.tolist())
        for d in diff:
            assert abs(best[d] - os_[-1]) <= 2e-5
        # ties broken by ascending index where the scores are exactly equal
        for j in range(1, n[i]):
            if score[i, j] == score[i, j - 1]:
                assert doc[i, j] > doc[i, j - 1]
        # arg-max chunk row
        for j in range(n[i]):
            d = doc[i, j]
            c = chunk[i, j]
            lo, hi = doc_off[d], doc_off[d + 1]
            if mc:
                hi = min(hi, lo + mc)
            assert lo <= c < hi
            cos_c = float(mods["rerank_ref"].cosine_f32(q[i], emb[c:c + 1])[0])
            assert abs(cos_c - best[d]) <= 2e-5


@pytest.mark.parametrize("layout,variant", [(0, 0), (1, 0), (0, 2), (1, 2), (0, 7), (1, 7), (0, 14), (0, 15)])
def test_dense_scan_vs_oracle(mods, layout, variant):
    rng = np.random.default_rng(17 + layout)
    doc_off, emb = _rand_chunked(rng, 700, 9, big=((5, 70), (300, 300), (301, 0), (699, 33)))
    ix = mods["CorpusIndex"](doc_ids=np.arange(700, dtype=np.int64) * 2 + 11, doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=700)
    eng = mods["DeviceEngine"](ix, max_queries=32, max_k=200, scan_layout=layout, scan_variant=variant)
    # variants 0 (default: resolves to 14), 14, 15 run the K-split kernel for every count, 2 (exact f32) for 33..64
    # queries per sweep; 7 and the interleaved layout stay on the 32-query wave-streaming kernel
    for Q in (1, 5, 16, 17, 32, 40) + ((50, 64, 70) if layout == 0 and variant in (0, 2, 14, 15) else ()):
        q = (rng.standard_normal((Q, 768)) * rng.uniform(0.5, 9)).astype(np.float32)
        q[0] = emb[123] * 4.0                                  # an exact hit
        for (k, mc) in ((100, 0), (200, 10), (7, 3)) if Q <= 40 else ((100, 0), (9, 2)):
            got = eng.dense_topk(q, k=k, max_chunks_per_doc=mc)
            _check_dense(mods, eng, doc_off, emb, q, k, mc, got)
    eng.close()


@pytest.mark.parametrize("variant", [0, 2, 7])
def test_dense_tiny_and_ragged(mods, variant):
    rng = np.random.default_rng(2)
    for n_docs, max_ch in ((1, 1), (3, 2), (40, 1), (17, 40), (5000, 3)):
        doc_off, emb = _rand_chunked(rng, n_docs, max_ch)
        if doc_off[-1] == 0:
            continue
        ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                                 chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=n_docs)
        eng = mods["DeviceEngine"](ix, max_queries=4, max_k=64, scan_variant=variant)
        q = rng.standard_normal((3, 768)).astype(np.float32)
        q[1] = 0.0                                              # zero query: cosine 0 everywhere (sklearn)
        got = eng.dense_topk(q, k=64)
        _check_dense(mods, eng, doc_off, emb, q, 64, 0, got)
        eng.close()


def test_wide_sweep_needs_a_bounded_document_span(mods):
    """The 64-query K-split kernel keeps per-document maxima in an LDS ring of 128 documents; a corpus where 32
    consecutive rows span more documents than that (a long run of chunk-less documents) stays on the 32-query kernel.
    Both give the oracle's answer for a batch of 40."""
    rng = np.random.default_rng(31)
    for gap, width in ((0, 64), (300, 32)):
        n = rng.integers(1, 7, size=900)
        if gap:
            n[400:400 + gap] = 0
        doc_off = np.zeros(901, np.int64); doc_off[1:] = np.cumsum(n)
        emb = rng.standard_normal((int(doc_off[-1]), 768)).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        ix = mods["CorpusIndex"](doc_ids=np.arange(900, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                                 chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=900)
        eng = mods["DeviceEngine"](ix, max_queries=8, max_k=100)
        assert eng.scan_arith() == "f16x2" and eng.scan_width() == width
        q = rng.standard_normal((40, 768)).astype(np.float32)
        _check_dense(mods, eng, doc_off, emb, q, 100, 0, eng.dense_topk(q, k=100))
        eng.close()


def test_wide_sweep_with_leading_chunkless_documents(mods):
    """A corpus -- or a shard cut by CorpusIndex.shard -- that BEGINS with a run of chunk-less documents.  The bind-time
    check of the K-split kernel only sees the documents between consecutive rows, so the kernel itself must retire
    the leading all-chunk-less blocks before its first row (round-1 advisor finding: ring slots aliased, chunk-less
    documents came back with finite scores).  Checked on the 64-query f16x2 sweep, the 128-query bf16 sweep and on a
    shard that starts inside such a run."""
    rng = np.random.default_rng(41)
    n = rng.integers(1, 7, size=900)
    n[0:300] = 0
    doc_off = np.zeros(901, np.int64); doc_off[1:] = np.cumsum(n)
    emb = rng.standard_normal((int(doc_off[-1]), 768)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    ix = mods["CorpusIndex"](doc_ids=np.arange(900, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=900)
    eng = mods["DeviceEngine"](ix, max_queries=8, max_k=700)
    assert eng.scan_arith() == "f16x2" and eng.scan_width() == 64          # the K-split kernel does run
    q = rng.standard_normal((128, 768)).astype(np.float32)
    for Q in (40, 64):
        got = eng.dense_topk(q[:Q], k=700)
        _check_dense(mods, eng, doc_off, emb, q[:Q], 700, 0, got)
        doc, score, chunk, cnt = [x.cpu().numpy() for x in got]
        assert (cnt == 600).all() and (doc[:, :600] >= 300).all()          # no chunk-less document is ever returned
    eng.enable_bf16()
    d_b, s_b, c_b, n_b = [x.cpu().numpy() for x in eng.dense_topk_batched(q, k=100)]      # 128 queries: one bf16 sweep
    d_e, s_e, c_e, n_e = [x.cpu().numpy() for x in eng.dense_topk(q, k=100)]
    assert (n_b == 100).all() and (d_b >= 300).all()
    assert np.abs(s_b - s_e).max() <= 8e-6 and (d_b == d_e).mean() > 0.995
    eng.close()
    # a shard that begins inside a chunk-less run: 2 shards balanced by chunk count, then the same cut by hand
    n2 = rng.integers(1, 7, size=900)
    n2[430:700] = 0                                                         # shard 1 of 2 will start with part of it
    off2 = np.zeros(901, np.int64); off2[1:] = np.cumsum(n2)
    emb2 = rng.standard_normal((int(off2[-1]), 768)).astype(np.float32)
    emb2 /= np.linalg.norm(emb2, axis=1, keepdims=True)
    full = mods["CorpusIndex"](doc_ids=np.arange(900, dtype=np.int64), doc_off=off2.astype(np.int32),
                               chunk_ids=np.arange(off2[-1], dtype=np.int64), emb=emb2, total_docs=900)
    for r in range(2):
        sh = full.shard(r, 2)
        so = np.asarray(sh.doc_off, dtype=np.int64)
        se = np.asarray(sh.emb)
        e2 = mods["DeviceEngine"](sh, max_queries=8, max_k=100)
        _check_dense(mods, e2, so, se, q[:40], 100, 0, e2.dense_topk(q[:40], k=100))
        e2.close()


def test_default_kernel_matches_the_reference_on_unit_rows(mods):
    """cosine_unit.npz: unit-norm rows, documents of 2..12 chunks, 40 queries, per-chunk cosines produced by the
    reference's own get_new_similarity (reranker_api.py:273-287).  On such rows the engine's DEFAULT dense kernel is
    the f16x2-split K-split kernel; its per-document maxima must be within the PROVEN bound of the arithmetic
    (8e-6), not merely within the 1e-5 task tolerance.  Q = 1 runs the 32-query instance, Q = 40
    the 64-query instance."""
    z = np.load(os.path.join(G, "cosine_unit.npz"))
    E, qs, exp, doc_off = z["E"], z["q"], z["expected"], z["doc_off"].astype(np.int64)
    n = len(doc_off) - 1
    ix = mods["CorpusIndex"](doc_ids=np.arange(n, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=E, total_docs=n)
    eng = mods["DeviceEngine"](ix, max_queries=64, max_k=n)
    assert eng.scan_arith() == "f16x2" and eng.scan_width() == 64
    want = np.stack([np.maximum.reduceat(exp[i], doc_off[:-1]) for i in range(len(qs))])   # per-document max of the golden
    warg = np.stack([[doc_off[d] + int(np.argmax(exp[i, doc_off[d]:doc_off[d + 1]])) for d in range(n)]
                     for i in range(len(qs))])
    for sel in (slice(0, 1), slice(0, 40)):
        doc, score, chunk, cnt = [x.cpu().numpy() for x in eng.dense_topk(qs[sel], k=n)]
        for i in range(doc.shape[0]):
            assert cnt[i] == n
            got = np.empty(n, np.float32); got[doc[i]] = score[i]
            assert np.abs(got - want[i]).max() <= 8e-6
            garg = np.empty(n, np.int64); garg[doc[i]] = chunk[i]
            bad = garg != warg[i]                                   # arg-max may differ only between near-equal chunks
            assert np.all(np.abs(exp[i, garg[bad]] - exp[i, warg[i][bad]]) <= 1.6e-5)
    eng.close()


def test_scan_arithmetic_is_chosen_from_the_row_norms(mods):
    """Default engine: f16-split products for (near) unit-norm rows, the exact f32 MFMA kernel as soon as one row
    norm leaves [0.5, 2]; scan_variant = 2 forces the exact kernel."""
    rng = np.random.default_rng(8)
    doc_off, emb = _rand_chunked(rng, 400, 5)
    mk = lambda e, **kw: mods["DeviceEngine"](mods["CorpusIndex"](
        doc_ids=np.arange(400, dtype=np.int64), doc_off=doc_off.astype(np.int32),
        chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=e, total_docs=400), max_queries=4, max_k=10, **kw)
    a = mk(emb); assert a.scan_arith() == "f16x2"
    b = mk(emb, scan_variant=2); assert b.scan_arith() == "f32"
    big = emb.copy(); big[7] *= 2.5
    c = mk(big); assert c.scan_arith() == "f32"
    small = emb.copy(); small[9] *= 0.3
    d = mk(small); assert d.scan_arith() == "f32"
    zero = emb.copy(); zero[3] = 0.0                       # a zero row has "norm 1" (sklearn) and stays eligible
    z = mk(zero); assert z.scan_arith() == "f16x2"
    q = rng.standard_normal((2, 768)).astype(np.float32)
    sa, sb = a.dense_topk(q, k=10)[1].cpu().numpy(), b.dense_topk(q, k=10)[1].cpu().numpy()
    assert np.abs(sa - sb).max() <= 8e-6                    # the proven bound; typically ~1e-7
    for e in (a, b, c, d, z):
        e.close()


def test_dense_massive_ties(mods):
    """Every chunk identical => every document has the same cosine: the k lowest indices, in order."""
    n = 9000
    v = np.random.default_rng(1).standard_normal(768).astype(np.float32)
    v /= np.linalg.norm(v)
    E = np.tile(v, (n, 1))
    ix = mods["CorpusIndex"](doc_ids=np.arange(n, dtype=np.int64), doc_off=np.arange(n + 1, dtype=np.int32),
                             chunk_ids=np.arange(n, dtype=np.int64), emb=E, total_docs=n)
    eng = mods["DeviceEngine"](ix, max_queries=4, max_k=1000)
    doc, score, chunk, cnt = [x.cpu().numpy() for x in eng.dense_topk(np.stack([v * 3, -v]), k=1000)]
    assert cnt.tolist() == [1000, 1000]
    assert doc[0].tolist() == list(range(1000)) and doc[1].tolist() == list(range(1000))
    assert abs(score[0, 0] - 1.0) < 1e-5 and abs(score[1, 0] + 1.0) < 1e-5
    eng.close()


def test_dense_streaming_pass_overflow_takes_the_gated_sweeps(mods):
    """More than 64 queries on a corpus the streaming pass accepts, one of them with a tie group of 9000 documents (more
    entries than a query's candidate list holds): the pass raises the device-side gate of that query's 64-query slice and
    the sweeps queued behind it redo the slice -- bit for bit what a 64-query call gives, with the tie rule intact; the
    other slice keeps the pass' answer; a batch without such a query, on the same engine afterwards, takes the pass again."""
    rng = np.random.default_rng(8)
    n = 60_000
    E = rng.standard_normal((n, 768)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    v = rng.standard_normal(768).astype(np.float32); v /= np.linalg.norm(v)
    tied = np.sort(rng.choice(n, size=9000, replace=False))
    E[tied] = v
    ix = mods["CorpusIndex"](doc_ids=np.arange(n, dtype=np.int64), doc_off=np.arange(n + 1, dtype=np.int32),
                             chunk_ids=np.arange(n, dtype=np.int64), emb=E, total_docs=n)
    eng = mods["DeviceEngine"](ix, max_queries=128, max_k=100, rerank_max_docs=0)
    assert eng.scan_width() == 128
    q = rng.standard_normal((100, 768)).astype(np.float32)
    q[7] = 2.5 * v
    got = [x.cpu().numpy() for x in eng.dense_topk(q, k=100)]
    # the tie query sits in the first 64-query slice: that slice comes back from the sweeps (bit for bit a 64-query call),
    # the other slice from the pass (exact f32 cosines: within rounding of the sweeps' f16x2-split scores)
    lo = [x.cpu().numpy() for x in eng.dense_topk(q[:64], k=100)]
    hi = [x.cpu().numpy() for x in eng.dense_topk(q[64:], k=100)]
    assert all(np.array_equal(a[:64], b) for a, b in zip(got, lo))
    assert np.array_equal(got[3][64:], hi[3]) and np.abs(got[1][64:] - hi[1]).max() <= 1e-5
    assert (got[0][64:] == hi[0]).mean() > 0.99
    assert got[3][7] == 100 and got[0][7].tolist() == tied[:100].tolist() and np.all(np.abs(got[1][7] - 1.0) < 1e-5)
    q[7] = rng.standard_normal(768).astype(np.float32)                        # no tie group any more: the pass itself answers
    again = [x.cpu().numpy() for x in eng.dense_topk(q, k=100)]
    ref2 = [np.concatenate(p) for p in zip(*[[x.cpu().numpy() for x in eng.dense_topk(q[s:s + 50], k=100)] for s in (0, 50)])]
    assert np.array_equal(again[3], ref2[3]) and np.abs(again[1] - ref2[1]).max() <= 1e-5
    assert (again[0] == ref2[0]).mean() > 0.99
    eng.close()


def test_streaming_pass_last_tile_with_a_partly_filled_fragment(mods):
    """the last row tile ends inside a wave's FIRST 16-row fragment ((rows mod 32) in 1..16), so that wave's
    second fragment lies wholly behind the matrix and accumulates products of the clamped last row.  A query equal to the last
    row makes those phantom rows pass the emission threshold: they must be masked (an emitted row index >= n_rows is an
    out-of-bounds read of chunk_doc further down).  Also an INNER tile of that shape (a 200-chunk document forces short
    tiles): the rows behind it belong to the next tile and must not be emitted twice."""
    rng = np.random.default_rng(17)
    for tail in (9, 16, 1):
        n_docs = 256 * 250 + tail                                # one chunk per document: tiles of exactly 256 rows + the tail
        n = np.ones(n_docs, np.int64)
        n[1000] = 200; n[1001] = 73                              # ... and short inner tiles (73 = 2 x 32 + 9 rows)
        doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
        C = int(doc_off[-1])
        emb_t = torch.randn((C, 768), generator=torch.Generator().manual_seed(tail))
        emb_t /= emb_t.norm(dim=1, keepdim=True)
        emb_t[C - 1] *= 1.9                                      # un-normalised products of the phantom rows would be larger still
        emb = emb_t.numpy()
        ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                                 chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
        eng = mods["DeviceEngine"](ix, max_queries=128, max_k=100, rerank_max_docs=0)
        assert eng.scan_width() == 128
        q = rng.standard_normal((100, 768)).astype(np.float32)
        q[3] = emb[C - 1] * 2.0                                  # the last row of the matrix
        q[70] = emb[C - 1] + 0.1 * q[70] / np.linalg.norm(q[70])
        q[5] = emb[doc_off[1001] + 72] * 0.7                     # the last row of the short inner tile
        got = eng.dense_topk(q, k=100)
        _check_dense(mods, eng, doc_off, emb, q[[3, 5, 70, 99]], 100, 0, [x[[3, 5, 70, 99]] for x in got])
        g = [x.cpu().numpy() for x in got]
        assert g[0][3, 0] == n_docs - 1 and g[2][3, 0] == C - 1 and abs(g[1][3, 0] - 1.0) <= 1e-5
        sw = [np.concatenate(p) for p in zip(*[[x.cpu().numpy() for x in eng.dense_topk(q[s:s + 50], k=100)] for s in (0, 50)])]
        assert np.array_equal(g[3], sw[3]) and np.abs(g[1] - sw[1]).max() <= 1e-5 and (g[0] == sw[0]).mean() > 0.99
        eng.close()
        # the same on the 256-query kernel, which streams the fragment-order copy of the rows: there the rows behind a tile are
        # the copy's zero padding (every tile starts at a multiple of 16 rows) or the next tile's -- masked all the same
        eng = mods["DeviceEngine"](ix, max_queries=256, max_k=100, rerank_max_docs=0)
        assert eng.scan_width() == 256
        q2 = np.concatenate([q, rng.standard_normal((100, 768)).astype(np.float32)])
        got = eng.dense_topk(q2, k=100)
        assert eng.dense_path() == 256
        _check_dense(mods, eng, doc_off, emb, q2[[3, 5, 70, 99, 150]], 100, 0, [x[[3, 5, 70, 99, 150]] for x in got])
        g2 = [x.cpu().numpy() for x in got]
        assert g2[0][3, 0] == n_docs - 1 and g2[2][3, 0] == C - 1 and abs(g2[1][3, 0] - 1.0) <= 1e-5
        assert all(np.array_equal(a_[:100], b_) for a_, b_ in zip(g2, g))          # (exact f32 rescoring: the kernels agree bit for bit)
        owned_with_copy = eng.owned_bytes()
        assert eng.row_copy_state() == "built" and owned_with_copy > C * 768 * 4    # the handle owns the copy and says so
        assert eng.row_image_state() == "none"                                      # (256 queries per call: one group per launch)
        eng.close()
        # ... and with the copy declined (MSR_CFG_NO_ROW_COPY; also what a failed allocation of the copy falls back to): the
        # same kernel reads the caller's row-major matrix, clamping the row index at the end of the matrix -- the same bits
        eng = mods["DeviceEngine"](ix, max_queries=256, max_k=100, rerank_max_docs=0, row_copy=False)
        assert eng.scan_width() == 256 and eng.row_copy_state() == "declined"
        assert owned_with_copy - eng.owned_bytes() >= C * 768 * 4                    # ... and is that much lighter without it
        g3 = [x.cpu().numpy() for x in eng.dense_topk(q2, k=100)]
        assert eng.dense_path() == 256 and all(np.array_equal(a_, b_) for a_, b_ in zip(g3, g2))
        eng.close()


def test_streaming_pass_vs_oracle_and_query_groups(mods):
    """msr_dense_topk with more than 64 queries on a corpus the streaming pass accepts (60 k documents of 0..8 chunks, one of
    200; ~950 row tiles): (a) against the ORACLE for a few queries of a 100-query call -- an exact hit, near-duplicates of
    rows, random ones, a non-unit row in the corpus -- at k = 100 and k = 10; (b) an engine with room for 4 groups of 128
    queries per call (700 queries = 512 + 188: passes queued back to back, one finish) returns bit for bit what one group
    per call returns; (c) a batch with a ZERO query (every cosine 0: a tie group of the whole corpus) overflows the pass and
    comes back from the gated sweeps -- still the oracle's answer, and only for its own 64-query slice."""
    rng = np.random.default_rng(91)
    n_docs = 60000
    n = rng.integers(0, 9, size=n_docs)
    n[5] = 200
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb_t = torch.randn((C, 768), generator=torch.Generator().manual_seed(4))
    emb_t /= emb_t.norm(dim=1, keepdim=True)
    emb_t[321] *= 1.6
    emb = emb_t.numpy()
    ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
    eng = mods["DeviceEngine"](ix, max_queries=64, max_k=100, rerank_max_docs=0)
    assert eng.scan_width() == 128
    q = rng.standard_normal((700, 768)).astype(np.float32) * rng.uniform(0.5, 12, size=(700, 1)).astype(np.float32)
    q[0] = emb[4567] * 3.0
    q[2:40] = emb[rng.integers(0, C, 38)] + 0.4 * q[2:40] / np.linalg.norm(q[2:40], axis=1, keepdims=True)
    pick = [0, 1, 2, 3, 40, 99]
    for k in (100, 10):
        got = eng.dense_topk(q[:100], k=k)
        _check_dense(mods, eng, doc_off, emb, q[pick], k, 0, [x[pick] for x in got])
    # the pass returns exact f32 cosines: they differ from the sweeps' f16x2-split scores in the last bits (a sweep-served
    # answer would be bit-equal to a 50-query call), so this also shows that the pass itself answered
    sweep = eng.dense_topk(q[:50], k=100)[1].cpu().numpy()
    assert not np.array_equal(eng.dense_topk(q[:100], k=100)[1].cpu().numpy()[:50], sweep)
    one = [x.cpu().numpy() for x in eng.dense_topk(q, k=100)]
    eng4 = mods["DeviceEngine"](ix, max_queries=512, max_k=100, rerank_max_docs=0)
    many = [x.cpu().numpy() for x in eng4.dense_topk(q, k=100)]
    for r in range(0, 640, 128):                             # (the last 60 queries of `one` are a sweep call: compared below)
        assert all(np.array_equal(a_[r:r + 128], b_[r:r + 128]) for a_, b_ in zip(one, many)), r
    assert np.array_equal(one[3], many[3]) and np.abs(one[1] - many[1]).max() <= 1e-6
    eng4.close()
    # room for 1024 queries per call: the 700 queries are THREE groups of 256 that share the rows of one launch (an odd number
    # of groups per XCD: some workgroups of the launch stay idle)
    eng8 = mods["DeviceEngine"](ix, max_queries=1024, max_k=100, rerank_max_docs=0)
    assert eng8.scan_width() == 256
    many = [x.cpu().numpy() for x in eng8.dense_topk(q, k=100)]
    assert eng8.dense_path() == 256
    for r in range(0, 640, 128):
        assert all(np.array_equal(a_[r:r + 128], b_[r:r + 128]) for a_, b_ in zip(one, many)), r
    assert np.array_equal(one[3], many[3]) and np.abs(one[1] - many[1]).max() <= 1e-6
    eng8.close()
    # (c) the zero query
    qz = q[:100].copy(); qz[1] = 0.0
    got = eng.dense_topk(qz, k=100)
    _check_dense(mods, eng, doc_off, emb, qz[[0, 1, 2]], 100, 0, [x[[0, 1, 2]] for x in got])
    # only the zero query's 64-query slice went back to the sweeps (one gate word per slice): rows 0..63 are what a 64-query
    # sweep call returns, rows 64..99 what the pass returned for the batch without the zero query
    gz = [x.cpu().numpy() for x in got]
    sw = [x.cpu().numpy() for x in eng.dense_topk(qz[:64], k=100)]
    ps = [x.cpu().numpy() for x in eng.dense_topk(q[:100], k=100)]
    assert all(np.array_equal(a_[:64], b_) for a_, b_ in zip(gz, sw))
    assert all(np.array_equal(a_[64:], b_[64:]) for a_, b_ in zip(gz, ps))
    eng.close()


def test_streaming_finish_rescores_the_emitted_rows_only(mods):
    """The streaming pass' finish recomputes in f32 only the rows the pass emitted for a candidate document (not all of its
    rows): the scores, the order and the FIRST arg-max row must still be the oracle's.  Documents built to stress that: one
    with two identical best rows (rows 1 and 3 of 5: the lower one wins), one whose best row is its last, one of 200 rows with
    a single good row in the middle; queries are those rows, scaled; both kernel widths (128 and 256 queries per pass), a
    batch of 300 on an engine that puts two query groups into one launch (the f16 image), and the bf16-candidate path."""
    rng = np.random.default_rng(17)
    n_docs = 40000
    n = rng.integers(1, 8, size=n_docs)
    n[100] = 5; n[200] = 4; n[300] = 200
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb = rng.standard_normal((C, 768)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    a, b, c = int(doc_off[100]), int(doc_off[200]), int(doc_off[300])
    emb[a + 3] = emb[a + 1]                                   # an exact tie inside document 100
    ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
    q = rng.standard_normal((300, 768)).astype(np.float32)
    q[0] = emb[a + 1] * 2.5                                   # both copies score 1.0
    q[1] = emb[b + 3] * 0.7                                   # the last row of document 200
    q[2] = emb[c + 117] * 4.0                                 # one row of the 200-row document
    q[3] = emb[a + 1] + 0.3 * q[3] / np.linalg.norm(q[3])     # near the tied pair: still tied with each other
    pick = [0, 1, 2, 3, 50, 99]
    for max_q, Q, width in ((64, 100, 128), (256, 200, 256), (512, 300, 256)):
        eng = mods["DeviceEngine"](ix, max_queries=max_q, max_k=100, rerank_max_docs=0)
        assert eng.scan_width() == width
        got = eng.dense_topk(q[:Q], k=100)
        assert eng.dense_path() == width
        _check_dense(mods, eng, doc_off, emb, q[pick], 100, 0, [x[pick] for x in got])
        doc, score, chunk, _ = [x.cpu().numpy() for x in got]
        assert doc[0, 0] == 100 and chunk[0, 0] == a + 1 and abs(score[0, 0] - 1.0) <= 1e-6      # first arg-max of the tie
        assert doc[1, 0] == 200 and chunk[1, 0] == b + 3
        assert doc[2, 0] == 300 and chunk[2, 0] == c + 117
        assert doc[3, 0] == 100 and chunk[3, 0] == a + 1
        if max_q == 512:
            # the batched path (bf16 candidates, K5) finishes the same way: the emitted rows of each candidate, in exact f32
            eng.enable_bf16()
            assert eng.batch_gemm_ok()
            got = eng.dense_topk_batched(q[:Q], k=100)
            _check_dense(mods, eng, doc_off, emb, q[pick], 100, 0, [x[pick] for x in got])
            doc, score, chunk, _ = [x.cpu().numpy() for x in got]
            assert [int(doc[i, 0]) for i in range(4)] == [100, 200, 300, 100]
            assert [int(chunk[i, 0]) for i in range(4)] == [a + 1, b + 3, c + 117, a + 1]
        eng.close()


def test_dense_cosine_golden(mods):
    """One chunk per document: the engine's scores are the reference's cosine_similarity values."""
    z = np.load(os.path.join(G, "cosine.npz"))
    E, q, exp = z["E"], z["q"], z["expected"]
    n = len(E)
    ix = mods["CorpusIndex"](doc_ids=np.arange(n, dtype=np.int64), doc_off=np.arange(n + 1, dtype=np.int32),
                             chunk_ids=np.arange(n, dtype=np.int64), emb=E, total_docs=n)
    for layout in (0, 1):
        eng = mods["DeviceEngine"](ix, max_queries=4, max_k=1000, scan_layout=layout)
        doc, score, chunk, cnt = [x.cpu().numpy() for x in eng.dense_topk(q[None, :], k=1000)]
        assert cnt[0] == n
        got = np.empty(n, np.float32); got[doc[0]] = score[0]
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-5)
        assert chunk[0].tolist() == doc[0].tolist()
        eng.close()


def test_batched_gemm_path_equals_exact_scan(mods):
    """More than 128 queries: msr_dense_topk_bf16 runs the tiled matrix-core GEMM (csrc/msr_gemm.hip: sample pass, emit
    pass, per-document maxima, exact f32 rescoring).  Its top-k must be the exact scan's: same scores to f32 rounding,
    same documents except where two scores are within rounding of each other.  Covers: query counts that do and do not
    fill the 256-query tiles (1, 2 and 3 tiles), k = 10 (sample stride > 1) and k = 100 (every tile sampled), chunk-less
    documents, an exact hit, a zero query; a repeated call returns identical results."""
    rng = np.random.default_rng(77)
    n_docs = 60000
    n = rng.integers(0, 9, size=n_docs)                    # 0..8 chunks: chunk-less documents included
    n[1000:1300] = 0
    n[5] = 200                                             # a long document, still inside one 256-row tile
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb_t = torch.randn((C, 768), generator=torch.Generator().manual_seed(3))
    emb_t /= emb_t.norm(dim=1, keepdim=True)
    emb_t[123] *= 1.7                                      # rows need not be unit norm (range [0.5, 2] keeps the default path)
    emb = emb_t.numpy()
    ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
    eng = mods["DeviceEngine"](ix, max_queries=64, max_k=100, rerank_max_docs=0)
    eng.enable_bf16()
    assert eng.batch_gemm_ok()
    q = rng.standard_normal((700, 768)).astype(np.float32) * rng.uniform(0.5, 12, size=(700, 1)).astype(np.float32)
    q[0] = emb[4567] * 3.0                                 # an exact hit
    q[1] = 0.0                                             # zero query: cosine 0 everywhere
    q[2:40] = emb[rng.integers(0, C, 38)] + 0.4 * q[2:40] / np.linalg.norm(q[2:40], axis=1, keepdims=True)
    for Q, k in ((129, 100), (300, 10), (700, 100), (512, 100)):
        exact = [x.cpu().numpy() for x in eng.dense_topk(q[:Q], k=k)]
        for rep in range(2 if Q == 300 else 1):
            got = [x.cpu().numpy() for x in eng.dense_topk_batched(q[:Q], k=k)]
            if rep:
                assert all(np.array_equal(a_, b_) for a_, b_ in zip(got, first))
            first = got
            assert np.array_equal(got[3], exact[3])
            assert np.abs(got[1] - exact[1]).max() <= 2e-6
            same = got[0] == exact[0]
            assert same.mean() > 0.99
            assert np.all(same | (np.abs(got[1] - exact[1]) <= 2e-6))
            # where the documents agree the arg-max chunk agrees too, unless two chunks of the document tie within rounding
            agree = (got[2] == exact[2]) | ~same
            assert agree.mean() > 0.999
            # ... and against the CPU oracle (oracle/dense_ref.py: sklearn's f32 cosine, per-document max, stable order), not
            # only against the engine's own exact path: queries of every 256-group the call holds
            for i in sorted({0, 2, 39, 128, Q - 1, Q // 2, min(Q - 1, 256), min(Q - 1, 511), min(Q - 1, 600)}):
                if i == 1:
                    continue
                oi, os_, oa = mods["dense_ref"].quick_search(emb, doc_off, q[i], k)
                assert got[3][i] == len(oi)
                np.testing.assert_allclose(got[1][i], os_, rtol=0, atol=1e-5)
                ok = (got[0][i] == oi) | (np.abs(np.r_[np.diff(os_), 1.0]) <= 4e-6) | (np.abs(np.r_[1.0, np.diff(os_)]) <= 4e-6)
                assert ok.all()
    assert eng.lib.msr_tune(eng.handle, 1, 3) < 0           # the product library has no tuning keys
    # msr_dense_topk with room for several groups of 128 queries per call (max_queries = 512: the passes of up to 4 groups
    # are queued back to back and finished together, 700 queries = 512 + 188) returns what one group per call returns
    one = [x.cpu().numpy() for x in eng.dense_topk(q, k=100)]
    eng4 = mods["DeviceEngine"](ix, max_queries=512, max_k=100, rerank_max_docs=0)
    many = [x.cpu().numpy() for x in eng4.dense_topk(q, k=100)]
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(one, many))
    eng4.close()
    # an engine whose max_queries / 128 is odd (a chunk of 300 .. 384 queries rounded up to 512 > 384 and was
    # refused): the call is cut into whole 256-query groups that fit (256 + the rest)
    eng3 = mods["DeviceEngine"](ix, max_queries=384, max_k=100, rerank_max_docs=0)
    for Q in (300, 384, 700):
        odd = [x.cpu().numpy() for x in eng3.dense_topk(q[:Q], k=100)]
        assert all(np.array_equal(a_[:256], b_[:256]) for a_, b_ in zip(one, odd))      # exact f32 finish on both: same bits
        assert np.array_equal(one[3][:Q], odd[3]) and np.abs(one[1][:Q] - odd[1]).max() <= 1e-5
        assert np.all((one[0][:Q] == odd[0]) | (np.abs(one[1][:Q] - odd[1]) <= 1e-5)) and (one[0][:Q] == odd[0]).mean() > 0.99
    eng3.close()
    # zero query: every cosine is exactly 0 -> the k lowest-indexed documents that have chunks, in order
    z = [x.cpu().numpy() for x in eng.dense_topk_batched(q[:200], k=100)]
    has = np.nonzero(n > 0)[0][:100]
    assert z[3][1] == 100 and z[0][1].tolist() == has.tolist() and np.all(z[1][1] == 0.0)
    eng.close()


def test_batched_path_without_gemm_preconditions(mods):
    """A document longer than 256 chunks cannot live inside one row tile: the engine keeps batches of more than 128
    queries on the sweeps (msr_batch_gemm_ok = 0) and the result is still the exact scan's."""
    rng = np.random.default_rng(78)
    doc_off, emb = _rand_chunked(rng, 3000, 6, big=((7, 300),))
    ix = mods["CorpusIndex"](doc_ids=np.arange(3000, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=3000)
    eng = mods["DeviceEngine"](ix, max_queries=64, max_k=50, rerank_max_docs=0)
    eng.enable_bf16()
    assert not eng.batch_gemm_ok()
    q = rng.standard_normal((150, 768)).astype(np.float32)
    a = [x.cpu().numpy() for x in eng.dense_topk_batched(q, k=50)]
    b = [x.cpu().numpy() for x in eng.dense_topk(q, k=50)]
    assert np.array_equal(a[3], b[3]) and np.abs(a[1] - b[1]).max() <= 2e-6
    assert np.all((a[0] == b[0]) | (np.abs(a[1] - b[1]) <= 2e-6))
    eng.close()


# ------------------------------------------------------------------------------------------------ rerank
def _rerank_case_index(mods, c):
    z = np.load(os.path.join(G, f"rerank_{c['case']}.npz"))
    urls_db = {int(u[0]): (u[1], u[2], u[3]) for u in c["urls"]}
    chunks = list(zip(z["chunk_id"].tolist(), z["chunk_doc"].tolist()))
    emb = {int(cid): z["emb"][i] for i, cid in enumerate(z["chunk_id"])}
    doc_len = {int(u[0]): 1 for u in c["urls"]}
    ix = mods["CorpusIndex"].from_tables({}, doc_len, {}, 1.0, chunks=chunks, emb=emb, urls_db=urls_db)
    return ix, z["q"]


@pytest.mark.parametrize("layout", [0, 1])
def test_rerank_chain_golden(mods, layout):
    for c in _load("rerank_chain.json")["cases"]:
        ix, q = _rerank_case_index(mods, c)
        pos = {int(d): i for i, d in enumerate(ix.doc_ids)}
        eng = mods["DeviceEngine"](ix, max_queries=4, max_k=16, rerank_max_docs=256, scan_layout=layout)
        M = 256
        cand = np.full((2, M), -1, np.int32); bm = np.zeros((2, M), np.float64)
        ids = [int(d) for d in c["doc_ids"]]
        known = [(pos[d], s) for d, s in zip(ids, c["similarities"]) if d in pos]
        for r in range(2):                                       # same request twice: rows are independent
            cand[r, :len(known)] = [k[0] for k in known]
            bm[r, :len(known)] = [k[1] for k in known]
        n = np.array([len(known)] * 2, np.int32)
        out = [x.cpu().numpy() for x in eng.rerank(np.stack([q, q]), cand, bm, n)]
        doc, score, orig, chunk, cnt, rows = out
        pooled = c["stages"][-1]
        exp = {d: (ch, s, o) for d, ch, s, o in zip(pooled["doc_id"], pooled["chunk_id"], pooled["new_similarity"],
                                                    pooled["old_similarity"])}
        for r in range(2):
            assert cnt[r] == len(exp)
            assert rows[r] == c["response"]["div"]["total_documents"]
            got_ids = [int(ix.doc_ids[d]) for d in doc[r, :cnt[r]]]
            assert sorted(got_ids) == sorted(exp)
            assert np.all(np.diff(score[r, :cnt[r]]) <= 0)
            for j, d in enumerate(got_ids):
                ch, s, o = exp[d]
                assert abs(score[r, j] - s) < 5e-6, (d, score[r, j], s)
                assert abs(orig[r, j] - o) < 1e-12
                assert int(ix.chunk_ids[chunk[r, j]]) == ch
        eng.close()


# ------------------------------------------------------------------------------------------------ merge
@pytest.mark.parametrize("bits", [32, 64])
def test_merge_topk(mods, bits):
    rng = np.random.default_rng(9)
    ix = mods["CorpusIndex"](doc_ids=np.arange(4, dtype=np.int64), doc_len=np.ones(4, np.int32),
                             term_off=np.array([0, 1], np.int64), post_doc=np.zeros(1, np.int32),
                             post_tf=np.ones(1, np.int32), idf=np.ones(1, np.float32), avgdl=1.0, total_docs=4)
    eng = mods["DeviceEngine"](ix, max_queries=2, max_k=16)
    dt = np.float32 if bits == 32 else np.float64
    # Two kernels share the work (msr_topk.hip): the counting merge over the lists' prefixes above a cut, and the merge tree
    # (power-of-two list counts and lengths: odd part counts and ks exercise its phantom lists and tails) for the queries whose
    # prefixes do not fit.  "random": list lengths 0 .. k, many exact ties; "full": every list k long, scores alike (the
    # counting merge takes these); "skewed": one list holds nearly all of the best (left to the tree).
    for G_, Q, k, shape in ((2, 3, 10, "random"), (8, 5, 100, "random"), (8, 2, 1000, "random"), (1, 1, 1, "random"),
                            (3, 4, 100, "random"), (5, 2, 37, "random"), (6, 3, 1000, "random"), (8, 3, 1000, "full"),
                            (2, 2, 1000, "full"), (4, 3, 100, "full"), (8, 3, 1000, "skewed"), (3, 2, 700, "skewed"),
                            (16, 2, 500, "full"), (64, 2, 100, "full")):
        docs = np.full((G_, Q, k), -1, np.int32); sc = np.full((G_, Q, k), -np.inf, dt); ns = np.zeros((G_, Q), np.int32)
        for g in range(G_):
            for qi in range(Q):
                m = int(rng.integers(0, k + 1)) if shape == "random" else k
                if shape == "random":
                    s = np.sort(rng.integers(0, 50, size=m).astype(dt) / 7)[::-1]     # many exact ties
                elif shape == "full":
                    s = np.sort(rng.standard_normal(m).astype(dt))[::-1]
                    if qi == 1:
                        s = np.round(s, 1)                                            # ties across and inside the lists
                else:
                    s = np.sort((rng.standard_normal(m) + (6.0 if g == 1 else 0.0)).astype(dt))[::-1]
                d = rng.choice(np.arange(g * 100000, (g + 1) * 100000), size=m, replace=False).astype(np.int32)
                o = np.lexsort((d, -s.astype(np.float64)))
                docs[g, qi, :m], sc[g, qi, :m], ns[g, qi] = d[o], s[o], m
        t = lambda a: torch.as_tensor(a).cuda()
        od, os_, on = [x.cpu().numpy() for x in eng.merge_topk(t(docs), t(sc), t(ns), k)]
        for qi in range(Q):
            parts = [(docs[g, qi, :ns[g, qi]].astype(np.int64), sc[g, qi, :ns[g, qi]]) for g in range(G_)]
            ei, es = mods["dense_ref"].merge_topk(parts, k)
            assert on[qi] == len(ei)
            assert od[qi, :on[qi]].tolist() == ei.tolist()
            assert os_[qi, :on[qi]].tolist() == es.tolist()
    eng.close()


# ------------------------------------------------------------------------------------------------ facades
def _ranked_equal(got, exp, tol):
    assert len(got) == len(exp)
    gs = np.array([g[1] for g in got]); es = np.array([e[1] for e in exp])
    np.testing.assert_allclose(gs, es, rtol=0, atol=tol)
    i = 0
    while i < len(exp):
        j = i + 1
        while j < len(exp) and abs(es[j] - es[j - 1]) <= 2 * tol:
            j += 1
        assert sorted(g[0] for g in got[i:j]) == sorted(e[0] for e in exp[i:j]), (i, j)
        i = j


def test_reranker_facade_matches_reference_response(mods):
    from msretr.reranker import Reranker, RerankNotFound
    for c in _load("rerank_chain.json")["cases"]:
        ix, q = _rerank_case_index(mods, c)
        for div in (True, False):
            rr = Reranker(ix, config={"diversification": div}, max_queries=4, max_k=16, rerank_max_docs=256)
            resp = rr.rerank([str(d) for d in c["doc_ids"]], c["similarities"], query_embedding=q)
            exp = c["response"]["div" if div else "nodiv"]
            assert resp["total_documents"] == exp["total_documents"] and resp["total_windows"] == exp["total_windows"]
            _ranked_equal([(d["doc_id"], d["similarity_score"]) for d in resp["document_scores"]],
                          [(d["doc_id"], d["similarity_score"]) for d in exp["document_scores"]], 5e-6)
            emap = {d["doc_id"]: d for d in exp["document_scores"]}
            for d in resp["document_scores"]:
                e = emap[d["doc_id"]]
                assert (d["title"], d["url"]) == (e["title"], e["url"])
                assert d["most_relevant_window"]["window_index"] == e["window_index"]
                assert abs(d["original_similarity"] - e["original_similarity"]) < 1e-12
            assert [w["doc_id"] for w in resp["top_windows"]] == [d["doc_id"] for d in resp["document_scores"]][:100]
            if c["case"] == 0 and div:
                with pytest.raises(RerankNotFound):                  # HTTP 401 in the reference
                    rr.rerank(["999999", "888888"], [1.0, 0.5], query_embedding=q)
            rr.engine.close()


def _small_web_corpus(mods):
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    ix = synthetic_corpus(3000, n_chunks=13000, n_terms=2500, seed=31)
    N = ix.n_docs
    ix.vocab = {f"t{i}": i for i in range(ix.n_terms)}
    ix.vocab["tübingen"] = 0
    ids = ix.doc_ids.numpy()
    ix.urls = [f"https://site{int(d) % 41}.de/page/{int(d)}" + ("?s=1" if d % 17 == 0 else "") for d in ids]
    ix.urls[100] = ix.urls[200].split("?")[0] + "?dup=1"
    ix.urls[7] = None
    ix.titles = [None if d % 29 == 0 else f"Title {int(d)}" for d in ids]
    ix.texts = [f"text of {int(d)} " * (30 if d % 5 == 0 else 2) for d in ids]
    for i, u in enumerate(ix.urls):
        if u is None:
            ix.titles[i] = None; ix.texts[i] = None
    terms, qvec = synthetic_queries(ix, 6, seed=32)
    return ix, terms, qvec


def test_retriever_two_stage_matches_oracle_chain(mods):
    from msretr.retriever import Retriever
    ix, terms, qvec = _small_web_corpus(mods)
    rt = Retriever(indexer=ix, tokenizer=lambda s: s.split(), max_queries=8, max_k=1000, rerank_max_docs=1000)
    z = {k: getattr(ix, k).numpy() for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
    z["avgdl"] = ix.avgdl
    ids = ix.doc_ids.numpy()
    urls_db_bm = {int(d): (ix.titles[i], ix.texts[i]) for i, d in enumerate(ids) if ix.urls[i] is not None}
    urls_rr = {int(d): (ix.urls[i], ix.titles[i], ix.texts[i]) for i, d in enumerate(ids) if ix.urls[i] is not None}
    chunk_doc = np.repeat(ids, np.diff(ix.doc_off.numpy()))
    queries = [" ".join("tübingen" if t == 0 else f"t{t}" for t in tl if t != 0) for tl in terms]   # city is appended
    got = rt.search_batch(queries, query_embeddings=[q.numpy() for q in qvec])
    for qi, query in enumerate(queries):
        pq = mods["msretr"].preprocess_query(query)
        tids = [ix.vocab.get(t, -1) for t in pq.split()]
        stage1 = mods["bm25_ref"].search(z, tids, 1000, 0.0, urls_db_bm)
        if not stage1:
            assert got[qi] == []
            continue
        resp = mods["rerank_ref"].rerank(urls_rr, ix.chunk_ids.numpy(), chunk_doc, ix.emb.numpy(), qvec[qi].numpy(),
                                         [str(r["doc_id"]) for r in stage1], [r["score"] for r in stage1])
        _ranked_equal([(d["doc_id"], d["score"]) for d in got[qi]],
                      [(d["doc_id"], d["similarity_score"]) for d in resp["document_scores"]], 1e-5)
        assert [d["rank"] for d in got[qi]] == list(range(1, len(got[qi]) + 1))
        assert all(len(d["snippet"]) <= 203 for d in got[qi])
    lines = rt.batch_search([(str(i + 1), q) for i, q in enumerate(queries)], query_embeddings=[q.numpy() for q in qvec])
    assert all(re_line.count("\t") == 3 for re_line in (l["formatted_line"] for l in lines))
    # dense full-scan API
    qs = rt.quick_search(query_embedding=qvec[0].numpy(), top_k=10)
    oi, os_, oa = mods["dense_ref"].quick_search(ix.emb.numpy(), ix.doc_off.numpy(), qvec[0].numpy(), 10)
    assert [r["doc_id"] for r in qs] == [int(ids[i]) for i in oi]
    # chunk-level results (return_unique_docs=False): the top chunks by cosine, several per document allowed
    E, off = ix.emb.numpy(), ix.doc_off.numpy()
    qn = qvec[0].numpy() / np.linalg.norm(qvec[0].numpy())
    cos = (E @ qn) / np.linalg.norm(E, axis=1)
    order = np.lexsort((np.arange(len(cos)), -cos))[:10]
    cs = rt.quick_search(query_embedding=qvec[0].numpy(), top_k=10, return_unique_docs=False)
    assert len(cs) == 10 and [r["rank"] for r in cs] == list(range(1, 11))
    np.testing.assert_allclose([r["score"] for r in cs], cos[order], rtol=0, atol=1e-5)
    cid = ix.chunk_ids.numpy()
    for r, c in zip(cs, order):
        if abs(r["score"] - cos[c]) < 1e-7:                  # (ties within rounding may swap neighbours)
            assert r["chunk_id"] == int(cid[c]) and r["doc_id"] == int(ids[np.searchsorted(off, c, side="right") - 1])
    rt._chunk_engine.close()
    rt.engine.close()


def test_sharded_on_one_gpu_equals_unsharded(mods):
    """Three shard engines on the same device + msr_merge_topk / bit-OR of the gather arrays == one engine."""
    ix, terms, qvec = _small_web_corpus(mods)
    full = mods["DeviceEngine"](ix, max_queries=8, max_k=300, rerank_max_docs=300)
    tl = [ix.term_ids(t) for t in terms]
    fb = [x.cpu() for x in full.bm25_topk(tl, k=300)]
    fd = [x.cpu() for x in full.dense_topk(qvec, k=50)]
    fr = [x.cpu() for x in full.rerank(qvec, fb[0], fb[1], fb[2])]
    world = 3
    shards = [ix.shard(r, world) for r in range(world)]
    engs = [mods["DeviceEngine"](s, max_queries=8, max_k=300, rerank_max_docs=300) for s in shards]
    glob = lambda t, base: torch.where(t >= 0, t + base, t)
    pb = [e.bm25_topk(tl, k=300) for e in engs]
    pd_ = [e.dense_topk(qvec, k=50) for e in engs]
    mb = engs[0].merge_topk(torch.stack([glob(p[0], s.doc_base) for p, s in zip(pb, shards)]),
                            torch.stack([p[1] for p in pb]), torch.stack([p[2] for p in pb]), 300)
    md = engs[1].merge_topk(torch.stack([glob(p[0], s.doc_base) for p, s in zip(pd_, shards)]),
                            torch.stack([p[1] for p in pd_]), torch.stack([p[3] for p in pd_]), 50)
    for a, b in zip(mb, fb):
        assert torch.equal(a.cpu(), b)                            # bitwise float64
    for a, b in zip(md, (fd[0], fd[1], fd[3])):
        assert torch.equal(a.cpu(), b)
    parts = [e.rerank_gather(qvec, mb[0], mb[2], doc_base=s.doc_base, row_base=s.row_base) for e, s in zip(engs, shards)]
    cos = parts[0][0].view(torch.int32)
    meta = parts[0][1]
    for c, m in parts[1:]:
        cos = cos | c.view(torch.int32); meta = meta | m
    mr = engs[2].rerank_fuse(mb[0], mb[1], mb[2], cos.view(torch.float32), meta)
    for a, b in zip(mr, fr):
        assert torch.equal(a.cpu(), b)
    for e in engs + [full]:
        e.close()


# ------------------------------------------------------------------------------------------------ batched bf16 path
def test_dense_batched_bf16_matches_exact_path(mods):
    rng = np.random.default_rng(41)
    doc_off, emb = _rand_chunked(rng, 5000, 9, big=((5, 70), (300, 300), (301, 0)))
    # make the top of the ranking non-trivial: plant near-duplicates of the queries
    ix = mods["CorpusIndex"](doc_ids=np.arange(5000, dtype=np.int64) * 2 + 11, doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=5000)
    eng = mods["DeviceEngine"](ix, max_queries=32, max_k=200)
    eng.enable_bf16()
    assert eng.batch_width() == 128                             # this corpus admits the 128-query sweep (64-document ring)
    for Q in (1, 16, 33, 64, 70, 128, 150):
        q = (rng.standard_normal((Q, 768)) * rng.uniform(0.5, 9)).astype(np.float32)
        q[0] = emb[123] * 4.0 + 0.05 * rng.standard_normal(768).astype(np.float32)
        for (k, mc) in ((100, 0), (200, 10), (7, 3)) if Q <= 70 else ((100, 0),):
            got = eng.dense_topk_batched(q, k=k, max_chunks_per_doc=mc)
            _check_dense(mods, eng, doc_off, emb, q, k, mc, got)          # vs the oracle, 1e-5
            ex = eng.dense_topk(q, k=k, max_chunks_per_doc=mc)
            assert torch.equal(got[3], ex[3])
            assert float((got[1] - ex[1]).abs().max()) <= 2e-6              # f32 rescore vs f32 MFMA scan
            same = (got[0] == ex[0]) | ((got[1] - ex[1]).abs() <= 2e-6)
            assert bool(same.all())
    eng.close()


def test_dense_batched_overflow_falls_back_to_exact(mods):
    """9000 identical documents: every one is a candidate (> 4096) => the call reports overflow and the host
    reruns the query on the exact f32 scan; the tie rule still holds."""
    n = 9000
    v = np.random.default_rng(1).standard_normal(768).astype(np.float32)
    v /= np.linalg.norm(v)
    ix = mods["CorpusIndex"](doc_ids=np.arange(n, dtype=np.int64), doc_off=np.arange(n + 1, dtype=np.int32),
                             chunk_ids=np.arange(n, dtype=np.int64), emb=np.tile(v, (n, 1)), total_docs=n)
    eng = mods["DeviceEngine"](ix, max_queries=32, max_k=100)
    eng.enable_bf16()
    doc, score, chunk, cnt = [x.cpu().numpy() for x in eng.dense_topk_batched(np.stack([v * 2, -v]), k=100)]
    assert cnt.tolist() == [100, 100] and doc[0].tolist() == list(range(100)) and doc[1].tolist() == list(range(100))
    eng.close()


def test_engine_from_mapped_snapshot_streams_to_hbm(mods, tmp_path, monkeypatch):
    """snapshot directory -> memory map -> pinned double buffer -> HBM gives the same engine as the
    in-memory index (BM25 bitwise, dense scores bitwise: same arrays, same kernels)."""
    import msretr.engine as me
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    ix = synthetic_corpus(3000, n_chunks=12000, n_terms=2000, device="cpu")
    d = str(tmp_path / "snap")
    ix.save_dir(d)
    back = mods["CorpusIndex"].load_dir(d, mmap=True)
    monkeypatch.setattr(me, "STREAM_MIN_BYTES", 1 << 16)        # stream everything larger than 64 KB ...
    calls = []
    real = me.stream_to_device
    monkeypatch.setattr(me, "stream_to_device", lambda a, dev, block_bytes=1 << 20: calls.append(a.shape) or real(a, dev, block_bytes))
    a = mods["DeviceEngine"](ix, max_queries=8, max_k=100)
    b = mods["DeviceEngine"](back, max_queries=8, max_k=100)
    assert (12000, 768) in calls                                 # ... in 1 MB blocks: the embeddings took 36 of them
    terms, qv = synthetic_queries(ix, 6, seed=3)
    ra, rb = a.bm25_topk([ix.term_ids(t) for t in terms], k=100), b.bm25_topk([back.term_ids(t) for t in terms], k=100)
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    da, db = a.dense_topk(qv, k=50), b.dense_topk(qv, k=50)
    for x, y in zip(da, db):
        assert torch.equal(x, y)
    a.close(); b.close()


def test_index_built_on_the_gpu_serves_the_same_results(mods):
    """token-id streams -> BM25 tables on the GPU (sort-based builder) == the same builder on the CPU
    device, and the engine bound to the GPU-built tables answers like the oracle on the CPU-built ones."""
    import time
    from msretr.index_build import bm25_index_from_token_ids
    rng = np.random.default_rng(77)
    n, V = 20000, 5000
    lens = rng.integers(0, 120, size=n)
    tok_off = np.zeros(n + 1, np.int64); tok_off[1:] = np.cumsum(lens)
    tok = (rng.zipf(1.2, size=int(tok_off[-1])) % V).astype(np.int32)
    doc_ids = np.arange(n, dtype=np.int64) * 3 + 7
    t0 = time.time(); host = bm25_index_from_token_ids(doc_ids, tok_off, tok, V, device="cpu"); t_cpu = time.time() - t0
    t0 = time.time(); dev = bm25_index_from_token_ids(doc_ids, tok_off, tok, V, device="cuda"); torch.cuda.synchronize()
    t_gpu = time.time() - t0
    print(f"index build, {int(tok_off[-1])} tokens: cpu device {t_cpu:.2f} s, gpu {t_gpu:.2f} s")
    for name in ("doc_len", "term_off", "post_doc", "post_tf", "idf"):
        assert np.array_equal(getattr(dev, name).cpu().numpy(), np.asarray(getattr(host, name))), name
    assert dev.avgdl == host.avgdl and dev.total_docs == host.total_docs
    eng = mods["DeviceEngine"](dev, max_queries=4, max_k=50)
    ref = {k: np.asarray(getattr(host, k)) for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
    ref.update(avgdl=host.avgdl)
    qs = [[17, 300, 17], [4999, 2], [1234]]
    got = [x.cpu().numpy() for x in eng.bm25_topk(qs, k=50)]
    for i, q in enumerate(qs):
        d, s_ = mods["bm25_ref"].topk(ref, q, 50, 0.0)
        assert got[0][i, :got[2][i]].tolist() == d.tolist() and got[1][i, :got[2][i]].tolist() == s_.tolist()
    eng.close()


# ------------------------------------------------------------------------------------------------ ABI robustness
def test_bind_rejects_malformed_index(mods):
    """The scoring kernel indexes LDS with the posting's document index: a malformed CSR must be refused at bind."""
    from msretr._abi import MsrError
    N = 100
    good = dict(doc_ids=np.arange(N, dtype=np.int64), doc_len=np.full(N, 5, np.int32),
                term_off=np.array([0, 3, 5], np.int64), post_doc=np.array([1, 4, 9, 0, 7], np.int32),
                post_tf=np.ones(5, np.int32), idf=np.ones(2, np.float32), avgdl=5.0, total_docs=N)
    mods["DeviceEngine"](mods["CorpusIndex"](**good), max_queries=2, max_k=4).close()
    bad = [("post_doc", np.array([1, 4, 4, 0, 7], np.int32), "ascending"),
           ("post_doc", np.array([1, 4, 100, 0, 7], np.int32), "outside"),
           ("post_doc", np.array([9, 4, 1, 0, 7], np.int32), "ascending"),
           ("term_off", np.array([0, 4, 3], np.int64), "offset"),
           ("post_tf", np.array([1, 0, 1, 1, 1], np.int32), "frequency"),
           ("doc_len", np.full(N, -1, np.int32), "doc_len")]
    for key, val, msg in bad:
        kw = dict(good); kw[key] = val
        with pytest.raises(MsrError) as ei:
            mods["DeviceEngine"](mods["CorpusIndex"](**kw), max_queries=2, max_k=4)
        assert msg in str(ei.value), (key, str(ei.value))
    # chunk side: non-monotone doc_off / wrong total
    emb = np.zeros((6, 768), np.float32)
    for off in ([0, 4, 2, 6], [0, 2, 4, 5], [1, 2, 4, 6]):
        ix = mods["CorpusIndex"](doc_ids=np.arange(3, dtype=np.int64), doc_off=np.array(off, np.int32),
                                 chunk_ids=np.arange(6, dtype=np.int64), emb=emb, total_docs=3)
        with pytest.raises(MsrError):
            mods["DeviceEngine"](ix, max_queries=2, max_k=4)


def test_queries_with_nan_or_zero_vectors(mods):
    rng = np.random.default_rng(3)
    doc_off, emb = _rand_chunked(rng, 300, 4)
    ix = mods["CorpusIndex"](doc_ids=np.arange(300, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=300)
    eng = mods["DeviceEngine"](ix, max_queries=8, max_k=10)
    q = rng.standard_normal((3, 768)).astype(np.float32)
    q[1, 5] = np.nan                                   # a poisoned query must not disturb its neighbours
    doc, score, chunk, n = [x.cpu().numpy() for x in eng.dense_topk(q, k=10)]
    assert n[1] == 0 and n[0] == 10 and n[2] == 10
    oi, os_, _ = mods["dense_ref"].quick_search(emb, doc_off, q[2], 10)
    assert doc[2].tolist() == oi.tolist()
    eng.close()


# ------------------------------------------------------------------------------------------------ queries.txt end to end
def test_queries_txt_end_to_end_matches_oracle(mods, tmp_path):
    """The reference's batch path (search_api.py:204-367) on a synthetic crawl: preprocess_query -> tokens ->
    BM25 top-1000 -> rerank -> top-100 -> `qnum<TAB>rank<TAB>url<TAB>score`; every line equals the oracle's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("demo", os.path.join(os.path.dirname(__file__), "..", "examples", "run_queries_txt.py"))
    demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
    from msretr.retriever import Retriever
    from msretr.text import preprocess_query, simple_tokenize
    ix = demo.synthetic_crawl(n_docs=1500)
    enc = demo.fake_encoder()
    rt = Retriever(embedder=enc, indexer=ix, tokenizer=simple_tokenize, max_queries=8, max_k=1000)
    qf = tmp_path / "queries.txt"
    qf.write_text("".join(f"{i + 1}\t{q}\n" for i, q in enumerate(demo.DEFAULT_QUERIES)), encoding="utf-8")
    out = tmp_path / "batch_search_results.txt"
    n = rt.batch_search_to_file(str(qf), str(out))
    lines = out.read_text(encoding="utf-8").splitlines()
    assert n == len(lines) and n > 0
    z = dict(doc_ids=ix.doc_ids, doc_len=ix.doc_len, term_off=ix.term_off, post_doc=ix.post_doc, post_tf=ix.post_tf,
             idf=ix.idf, avgdl=ix.avgdl)
    urls_bm = {int(d): (ix.titles[i], ix.texts[i]) for i, d in enumerate(ix.doc_ids)}
    urls_rr = {int(d): (ix.urls[i], ix.titles[i], ix.texts[i]) for i, d in enumerate(ix.doc_ids)}
    chunk_doc = np.repeat(ix.doc_ids, np.diff(ix.doc_off))
    exp = []
    for qi, q in enumerate(demo.DEFAULT_QUERIES):
        pq = preprocess_query(q)
        tids = [ix.vocab.get(t, -1) for t in simple_tokenize(pq)]
        s1 = mods["bm25_ref"].search(z, tids, 1000, 0.0, urls_bm)
        if not s1:
            continue
        resp = mods["rerank_ref"].rerank(urls_rr, ix.chunk_ids, chunk_doc, ix.emb, enc(pq),
                                         [str(r["doc_id"]) for r in s1], [r["score"] for r in s1])
        for rank, d in enumerate(resp["document_scores"], start=1):
            exp.append((str(qi + 1), rank, d["url"], d["similarity_score"]))
    assert len(lines) == len(exp)
    mism = 0
    for line, (qn, rank, url, score) in zip(lines, exp):
        a = line.split("\t")
        assert a[0] == qn and int(a[1]) == rank and abs(float(a[3]) - score) < 1.5e-3       # 3 decimals in the file
        mism += a[2] != url
    assert mism <= 2                       # a near-tie may swap two neighbours
    rt.engine.close()


def test_http_routes_on_the_gpu_retriever(mods, tmp_path):
    """server.py over a REAL Retriever (HIP kernels behind every route): /api/search shape, /api/batch_search and
    /api/batch_search_file (search_api.py:204-367) produce the same lines as Retriever.batch_search_to_file, /rerank
    returns the reranker's response for stage-1 output and 401 for unknown documents (reranker_api.py:348-349)."""
    import importlib.util
    from fastapi.testclient import TestClient
    spec = importlib.util.spec_from_file_location("demo", os.path.join(os.path.dirname(__file__), "..", "examples", "run_queries_txt.py"))
    demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
    from msretr.retriever import Retriever
    from msretr.server import create_app
    from msretr.text import simple_tokenize
    ix = demo.synthetic_crawl(n_docs=1200)
    rt = Retriever(embedder=demo.fake_encoder(), indexer=ix, tokenizer=simple_tokenize, max_queries=8, max_k=1000)
    qf = tmp_path / "queries.txt"
    qf.write_text("".join(f"{i + 1}\t{q}\n" for i, q in enumerate(demo.DEFAULT_QUERIES)), encoding="utf-8")
    direct = tmp_path / "direct.txt"
    n_direct = rt.batch_search_to_file(str(qf), str(direct))
    out = tmp_path / "batch_search_results.txt"
    c = TestClient(create_app(rt, queries_file=str(qf), results_file=str(out)))
    r = c.post("/api/search", json={"query": demo.DEFAULT_QUERIES[0], "top_k": 1000, "query_id": "q1"})
    assert r.status_code == 200
    docs = r.json()["documents"]
    assert 0 < len(docs) <= 100 and [d["rank"] for d in docs] == list(range(1, len(docs) + 1))
    assert set(docs[0]) == {"query_id", "rank", "url", "score", "title", "snippet", "domain", "doc_id"}
    assert all(docs[i]["score"] >= docs[i + 1]["score"] for i in range(len(docs) - 1))
    b = c.post("/api/batch_search").json()
    assert b["total_queries"] == len(demo.DEFAULT_QUERIES) and b["total_results"] == n_direct
    f = c.post("/api/batch_search_file").json()
    assert f["total_results"] == n_direct and f["output_file"] == str(out)
    assert out.read_text(encoding="utf-8") == direct.read_text(encoding="utf-8")
    assert [x["formatted_line"] for x in b["results"]] == direct.read_text(encoding="utf-8").splitlines()
    s1 = rt.bm25.search(demo.DEFAULT_QUERIES[0] + " tübingen", top_k=50)
    if s1:
        rr = c.post("/rerank", json={"doc_ids": [str(x["doc_id"]) for x in s1], "similarities": [x["score"] for x in s1],
                                      "query": demo.DEFAULT_QUERIES[0]})
        assert rr.status_code == 200 and set(rr.json()) >= {"document_scores", "top_windows", "total_documents", "total_windows"}
    assert c.post("/rerank", json={"doc_ids": ["999999999"], "similarities": [1.0], "query": "x"}).status_code == 401
    assert c.get("/api/health").json() == {"status": "healthy", "search_engine_ready": True}
    rt.engine.close()


def test_index_build_kernels_on_the_reference_fixture(mods):
    """msr_build_postings on the crawls of tests/golden/bm25_build.json: the tables equal what the reference's own
    BM25._process_document_batch returned for them (doc_stats, term_freq rows, doc_freq / total_freq per term)."""
    from msretr.index_build import bm25_index_from_token_ids, normalise_document_text
    for case in _load("bm25_build.json"):
        docs = [tuple(d) for d in case["documents"]]
        toks = [normalise_document_text(t, x).split() for _, t, x in docs]
        vocab = {}
        for tl in toks:
            for w in tl:
                vocab.setdefault(w, len(vocab))
        tok_off = np.zeros(len(docs) + 1, np.int64); tok_off[1:] = np.cumsum([len(t) for t in toks])
        tok_ids = np.array([vocab[w] for t in toks for w in t], np.int32)
        ix = bm25_index_from_token_ids([d for d, _, _ in docs], tok_off, tok_ids, len(vocab), device="cuda")
        ids, lens = ix.doc_ids.tolist(), ix.doc_len.cpu().numpy().tolist()
        assert dict(zip(ids, lens)) == {d: l for d, l in case["doc_stats"]} and ids == sorted(ids)
        toff, pd_, ptf = [x.cpu().numpy() for x in (ix.term_off, ix.post_doc, ix.post_tf)]
        got = {(ids[d], w): int(f) for w, i in vocab.items() for d, f in zip(pd_[toff[i]:toff[i + 1]], ptf[toff[i]:toff[i + 1]])}
        assert got == {(d, t): f for d, t, f in case["term_freq"]}, case["name"]
        for w, i in vocab.items():
            assert [int(toff[i + 1] - toff[i]), int(ptf[toff[i]:toff[i + 1]].sum())] == case["term_updates"][w]
            assert np.all(np.diff(pd_[toff[i]:toff[i + 1]]) > 0)
        assert set(vocab) == set(case["term_updates"]) and ix.total_docs == len(case["doc_stats"])


def test_build_postings_rejects_out_of_range_token_ids(mods):
    """the C entry point itself (not only its Python wrapper) refuses a token id outside [0, n_terms) -- the radix
    pass count and the doc_freq scatter index with it."""
    import ctypes as C

    from msretr import _abi
    lib = _abi.load()
    dev = torch.device("cuda", 0)
    off = torch.tensor([0, 3, 5], dtype=torch.int64, device=dev)
    term_off = torch.zeros(5, dtype=torch.int64, device=dev)
    n_post = C.c_int64(-1)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    for bad in ([0, 1, 4, 2, 3], [0, -1, 1, 2, 3]):
        tok = torch.tensor(bad, dtype=torch.int32, device=dev)
        rc = lib.msr_build_postings(ptr(off), ptr(tok), 2, 4, ptr(term_off), C.c_void_p(0), C.c_void_p(0), 0, C.byref(n_post), stream)
        assert rc == -1 and b"token id outside" in lib.msr_last_error(None)
    tok = torch.tensor([0, 1, 3, 2, 3], dtype=torch.int32, device=dev)
    rc = lib.msr_build_postings(ptr(off), ptr(tok), 2, 4, ptr(term_off), C.c_void_p(0), C.c_void_p(0), 0, C.byref(n_post), stream)
    assert rc == 0 and n_post.value == 5


def test_index_build_kernels_equal_the_reference_tables(mods):
    """msr_build_postings (per-document sort + run lengths, stable radix sort by term, boundary doc_freq --
    csrc/msr_build.hip) against the oracle's builder (oracle/build_ref.py, pinned to the output of the reference's own
    BM25._process_document_batch by tests/golden/bm25_build.json; bm25_indexer.py:196-243, 130-147, 346-369): postings,
    lengths, offsets, idf bits and avgdl identical.  Covers unsorted doc ids, token-less
    documents (no row), documents longer than one 4096-token chunk (their chunks' counts are merged), a term present
    in every document, and vocabularies that need 1, 2 and 3 radix passes."""
    from types import SimpleNamespace

    from msretr.index_build import bm25_index_from_token_ids
    from oracle import build_ref
    rng = np.random.default_rng(19)
    for n, V, long_docs, p_uni in ((300, 200, (), 0.3), (700, 40_000, (5, 77), 0.3),
                                   (400, 300_000, tuple(range(9, 400, 30)), 0.9)):
        doc_ids = rng.permutation(np.arange(1000, 1000 + 3 * n, 3))[:n]          # unsorted, with gaps
        lens = rng.integers(0, 120, size=n)
        lens[[3, 11]] = 0                                                         # token-less documents: no row
        for d in long_docs:
            lens[d] = int(rng.integers(5000, 13000))                             # 2..4 chunks
        toks = []
        for l in lens:
            t = np.minimum(rng.zipf(1.2, size=l), V - 1).astype(np.int64)
            m = rng.random(l) < p_uni
            t[m] = rng.integers(0, V, size=int(m.sum()))
            if l:
                t[0] = 0                                                          # term 0 in every document
            toks.append(t.tolist())
        ref = SimpleNamespace(**build_ref.index_from_tokens(doc_ids, [[f"w{t}" for t in tl] for tl in toks]))
        ids_of = {k: v for k, v in ref.vocab.items()}                             # the oracle's numbering
        assert (len(ref.vocab) > 65536) == (V == 300_000)                         # the last case needs a third radix pass
        tok_off = np.zeros(n + 1, np.int64); tok_off[1:] = np.cumsum(lens)
        tok_ids = np.array([ids_of[f"w{t}"] for tl in toks for t in tl], np.int32)
        got = bm25_index_from_token_ids(doc_ids, tok_off, tok_ids, len(ref.vocab), device="cuda")
        assert got.total_docs == ref.total_docs == int((lens > 0).sum()) and got.avgdl == ref.avgdl
        for name in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf"):
            a = getattr(got, name)
            a = a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
            assert np.array_equal(a, np.asarray(getattr(ref, name))), (name, n, V)
        assert np.array_equal(got.idf.cpu().numpy().view(np.uint32), ref.idf.view(np.uint32))
    # and the built index serves queries: BM25 top-k over it equals the oracle on the reference-shaped tables
    eng = mods["DeviceEngine"](got, max_queries=4, max_k=50, rerank_max_docs=0)
    z = {k: (getattr(got, k).cpu().numpy() if torch.is_tensor(getattr(got, k)) else np.asarray(getattr(got, k)))
         for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
    z["avgdl"] = got.avgdl
    qs = [[0, 5, 17], [3, 3, 250], [1]]
    doc, score, cnt = [x.cpu().numpy() for x in eng.bm25_topk(qs, k=50)]
    for i, t in enumerate(qs):
        oi, os_ = mods["bm25_ref"].topk(z, t, 50)
        assert doc[i, :cnt[i]].tolist() == oi.tolist() and score[i, :cnt[i]].tolist() == os_.tolist()
    eng.close()


def test_diversify_kernel_matches_reference(mods):
    """msr_diversify against the reference's hybrid_diversification: the executed fixture (tests/golden/diversification.json)
    and random lists -- many score ties, more "high" domains than top_k, lists shorter than top_k, rejected documents,
    diversification off -- against oracle.rerank_ref.hybrid_diversification (pinned to the same fixture).  Scores must be
    EQUAL as float64: the kernel performs the reference's operations (delta = first_dropped - last_kept + 1e-4, max(0, s - delta))."""
    import json
    from urllib.parse import urlparse
    rr = mods["rerank_ref"]
    ix = mods["CorpusIndex"](doc_ids=np.arange(5000, dtype=np.int64), doc_off=np.arange(5001, dtype=np.int32),
                             chunk_ids=np.arange(5000, dtype=np.int64),
                             emb=np.eye(768, dtype=np.float32)[np.arange(5000) % 768], total_docs=5000)
    eng = mods["DeviceEngine"](ix, max_queries=8, max_k=100, rerank_max_docs=1000)
    dev = eng.device

    def run(cases, top_k, diversification=True):
        """cases: list of lists of (doc, domain id or -1, score), each sorted by (score desc, doc asc)."""
        M = 1000
        Q = len(cases)
        doc = torch.full((Q, M), -1, dtype=torch.int32); score = torch.full((Q, M), -float("inf"), dtype=torch.float64)
        n = torch.zeros(Q, dtype=torch.int32)
        dom = np.full(5000, -1, np.int32)
        for q, c in enumerate(cases):
            n[q] = len(c)
            for j, (d, g, s_) in enumerate(c):
                doc[q, j] = d; score[q, j] = s_
                assert dom[d] in (-1, g) or g == -1
                dom[d] = g
        eng.bind_doc_domains(dom)
        orig = score.clone() * 0.5
        chunk = doc.clone()
        out = eng.diversify((doc.to(dev), score.to(dev), orig.to(dev), chunk.to(dev), n.to(dev)), top_k=top_k,
                            diversification=diversification)
        od, os_, oo, oc, on = [x.cpu().numpy() for x in out]
        res = []
        for q in range(Q):
            k = int(on[q])
            assert np.all(od[q, k:] == -1) and np.all(oc[q, :k] == od[q, :k])
            res.append([(int(od[q, j]), float(os_[q, j])) for j in range(k)])
        return res

    # (1) the reference-executed fixture
    with open(os.path.join(G, "diversification.json")) as f:
        fx = json.load(f)["cases"]
    for c in fx:
        doms = {}
        case = [(int(i), doms.setdefault(urlparse(u).netloc.lower(), len(doms)), float(s_)) for i, u, s_ in c["input"]]
        got = run([case], c["top_k"])[0]
        assert [[d, s_] for d, s_ in got] == c["expected"], (got, c["expected"])
    # (2) random lists against the oracle's restatement
    rng = np.random.default_rng(5)
    for trial in range(12):
        n_items = int(rng.choice([1, 7, 60, 99, 100, 101, 400, 1000]))
        n_dom = int(rng.choice([1, 3, 40, 150, 900]))
        top_k = int(rng.choice([5, 100]))
        hi_frac = float(rng.choice([0.0, 0.02, 0.3, 0.9]))
        docs = np.sort(rng.choice(5000, n_items, replace=False))
        sc = np.where(rng.random(n_items) < hi_frac, 0.8 + 0.2 * rng.random(n_items), 0.8 * rng.random(n_items))
        if trial % 2:
            sc = np.round(sc, 2)                                   # many exact ties, also across the two tiers' drops
        if trial == 3:
            sc[:] = 0.0                                            # last_kept = 0: the shifted tail clamps at 0
        order = np.lexsort((docs, -sc))
        docs, sc = docs[order], sc[order]
        dom = rng.integers(0, n_dom, n_items)
        rejected = rng.random(n_items) < (0.1 if trial % 3 == 0 else 0.0)
        case = [(int(d), -1 if r else int(g), float(s_)) for d, g, s_, r in zip(docs, dom, sc, rejected)]
        for div in (True, False):
            got = run([case], top_k, div)[0]
            items = [{"doc_id": d, "url": f"https://h{g}.de/x", "similarity_score": s_} for d, g, s_ in case if g >= 0]
            exp = rr.hybrid_diversification(items, top_k=top_k) if div else items[:top_k]
            assert got == [(e["doc_id"], e["similarity_score"]) for e in exp], (trial, div)
    eng.close()


def test_split_dense_call_with_a_cross_shard_bound_equals_unsharded(mods):
    """msr_dense_topk_begin / _end (the dense stage of a doc-sharded run): three shard engines on one GPU each vouch for
    ceil(k / 3) of their own documents, the minimum over the shards bounds the k-th cosine of the whole corpus from below, and
    every shard rescores only what can be in the global top-k.  The shards' lists are SHORTER than k; merged they are the
    unsharded engine's list bit for bit (documents, exact f32 scores, arg-max chunks) -- for planted near-duplicates, an exact
    hit, random directions, k = 100 and k = 10; a shard that holds none of a query's top documents contributes (almost) nothing."""
    rng = np.random.default_rng(123)
    n_docs = 90000
    n = rng.integers(1, 9, size=n_docs)
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb_t = torch.randn((C, 768), generator=torch.Generator().manual_seed(9))
    emb_t /= emb_t.norm(dim=1, keepdim=True)
    emb = emb_t.numpy()
    ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
    Q = 200
    q = rng.standard_normal((Q, 768)).astype(np.float32) * rng.uniform(0.5, 12, size=(Q, 1)).astype(np.float32)
    q[0] = emb[12345] * 3.0                                    # exact hit (first shard)
    q[1:60] = emb[rng.integers(0, C, 59)] + 0.4 * q[1:60] / np.linalg.norm(q[1:60], axis=1, keepdims=True)
    q[60] = emb[C - 1] * 0.5                                   # exact hit in the LAST shard's last tile
    full = mods["DeviceEngine"](ix, max_queries=256, max_k=100, rerank_max_docs=0)
    world = 3
    shards = [ix.shard(r, world) for r in range(world)]
    engs = [mods["DeviceEngine"](s_, max_queries=256, max_k=100, rerank_max_docs=0) for s_ in shards]
    glob = lambda t, base: torch.where(t >= 0, t + base, t)
    for k in (100, 10):
        ref = full.dense_topk(q, k=k)
        assert all(e.dense_split_max(k) >= Q for e in engs)
        parts = [e.dense_begin(q, k=k, k_part=(k + world - 1) // world) for e in engs]
        bound = torch.stack(parts).min(dim=0).values
        outs = [e.dense_end(Q, k=k, bound=bound) for e in engs]
        assert all(int(o[3].max()) <= k for o in outs) and sum(int(o[3].sum()) for o in outs) < 0.8 * world * Q * k   # shorter lists
        assert all(bool((o[3] >= 0).all()) for o in outs)
        # payload merge (the arg-max chunk travels with its document), as the sharded engine does after the all-gather
        docs = torch.stack([glob(o[0], s_.doc_base) for o, s_ in zip(outs, shards)])
        chunks = torch.stack([glob(o[2], s_.row_base) for o, s_ in zip(outs, shards)])
        m_doc, m_score, m_n = engs[0].merge_topk(docs, torch.stack([o[1] for o in outs]), torch.stack([o[3] for o in outs]), k)
        assert torch.equal(m_n, ref[3]) and torch.equal(m_doc, ref[0]) and torch.equal(m_score, ref[1])
        # the winning chunk of every merged document, looked up in the shard lists
        look = {}
        for g in range(world):
            d_, c_, n_ = docs[g].cpu().numpy(), chunks[g].cpu().numpy(), outs[g][3].cpu().numpy()
            for qi in (0, 1, 60, 199):
                for j in range(n_[qi]):
                    look[(qi, int(d_[qi, j]))] = int(c_[qi, j])
        rd, rc = ref[0].cpu().numpy(), ref[2].cpu().numpy()
        for qi in (0, 1, 60, 199):
            assert [look[(qi, int(d))] for d in rd[qi]] == rc[qi].tolist()
        # without a bound the halves are the plain call
        p0 = engs[0].dense_begin(q, k=k, k_part=k)
        plain = engs[0].dense_end(Q, k=k, bound=None)
        whole = engs[0].dense_topk(q, k=k)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(plain, whole))
    # misuse is refused, with a reason
    with pytest.raises(mods["msretr"].MsrError):
        engs[0].dense_end(Q, k=100)                             # no begin pending
    with pytest.raises(mods["msretr"].MsrError):
        engs[0].dense_begin(q[:10], k=100)                      # <= 64 queries: the sweeps, not the streaming pass
    engs[0].dense_begin(q, k=100)
    with pytest.raises(mods["msretr"].MsrError):
        engs[0].dense_topk(q, k=100)                            # the pending begin's scratch is in use
    with pytest.raises(mods["msretr"].MsrError):
        engs[0].dense_begin(q, k=100)
    assert int(engs[0].dense_end(Q, k=100)[3].min()) == 100     # ... and the pair still completes
    for e in engs + [full]:
        e.close()


def test_compact_rerank_exchange_equals_the_dense_halves(mods):
    """msr_rerank_plan / _gather_records / _scatter against msr_rerank_gather_blocks / _combine: three shard engines on one GPU,
    candidate lists with documents of every shard, absent documents (-1, out of range), short lists and an empty one; the
    all-to-all is done by hand from the plan's N x N matrix.  Every rank's (cos, meta) of ITS queries must equal the join of
    the dense halves bit for bit, the counts must equal a numpy count, and the records a rank receives must all be for its own
    queries."""
    from msretr.distributed import RECORD_WORDS, _RerankPlan
    rng = np.random.default_rng(77)
    n_docs = 30000
    n = rng.integers(1, 13, size=n_docs)                      # (some documents have more than 10 chunks: the first 10 take part)
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb = torch.randn((C, 768), generator=torch.Generator().manual_seed(4)).numpy()
    ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
    world, Q, M = 3, 11, 203                                   # (Q not a multiple of world, M not a multiple of 8)
    Qs = (Q + world - 1) // world
    shards = [ix.shard(r, world) for r in range(world)]
    engs = [mods["DeviceEngine"](s_, max_queries=16, max_k=16, rerank_max_docs=256) for s_ in shards]
    bounds = torch.tensor([s_.doc_base for s_ in shards] + [n_docs], dtype=torch.int32).cuda()
    cand = np.stack([rng.choice(n_docs, size=M, replace=False) for _ in range(Q)]).astype(np.int32)
    cand[0, :40] = np.arange(40) + shards[1].doc_base           # a run of one shard's documents (whole blocks of 8 owned)
    cand[1, ::3] = -1                                           # absent documents
    cand[2, 5] = n_docs + 7                                     # out of every shard's range
    cn = np.full(Q, M, np.int32); cn[3] = 17; cn[4] = 0; cn[5] = 1
    q = rng.standard_normal((Q, 768)).astype(np.float32) * 3
    cand_t, cn_t = torch.as_tensor(cand).cuda(), torch.as_tensor(cn).cuda()
    # the dense form, as distributed.py's a2a="blocks" runs it
    block = (Qs * M * 13 + 3) // 4 * 4
    sends = []
    for e, s_ in zip(engs, shards):
        buf = torch.zeros((world, block), dtype=torch.int32, device="cuda")
        e.rerank_gather_blocks(q, cand_t, cn_t, buf, Qs, doc_base=s_.doc_base, row_base=s_.row_base)
        sends.append(buf)
    # the compact form
    plans, recs = [], []
    for r, (e, s_) in enumerate(zip(engs, shards)):
        plan = _RerankPlan(world, Q, Qs, M, "cuda")
        e.rerank_plan(cand_t, cn_t, bounds, r, Qs, plan)
        rec = torch.full((Q * M * RECORD_WORDS,), -7, dtype=torch.int32, device="cuda")
        e.rerank_gather_records(q, cand_t, cn_t, plan, rec, doc_base=s_.doc_base, row_base=s_.row_base)
        plans.append(plan); recs.append(rec)
    torch.cuda.synchronize()
    own = np.full((Q, M), -1)
    b = bounds.cpu().numpy()
    for qi in range(Q):
        for m in range(cn[qi]):
            d = cand[qi, m]
            if b[0] <= d < b[-1]:
                own[qi, m] = np.searchsorted(b, d, side="right") - 1
    counts = np.stack([(own == s_).sum(axis=1) for s_ in range(world)])
    pair = np.array([[counts[s_, o * Qs:(o + 1) * Qs].sum() for o in range(world)] for s_ in range(world)])
    for plan in plans:
        assert np.array_equal(plan.counts.cpu().numpy(), counts) and np.array_equal(plan.pair.cpu().numpy(), pair)
    for r in range(world):
        lo, hi = min(Q, r * Qs), min(Q, (r + 1) * Qs)
        # what the all-to-all delivers to rank r: source g's records for r's queries, sources in order
        got = []
        for g in range(world):
            first = int(pair[g, :r].sum())
            got.append(recs[g][first * RECORD_WORDS:(first + int(pair[g, r])) * RECORD_WORDS])
        recv = torch.cat(got + [torch.zeros(8 * RECORD_WORDS, dtype=torch.int32, device="cuda")])
        rv = recv[:int(pair[:, r].sum()) * RECORD_WORDS].view(-1, RECORD_WORDS).cpu().numpy()
        assert ((rv[:, 14] >= lo) & (rv[:, 14] < hi)).all() and (rv[:, 0] >= 0).all() and (rv[:, 0] < M).all()
        cos, meta = engs[r].rerank_scatter(recv, plans[r], lo, hi - lo, M)
        parts = torch.stack([s_[r] for s_ in sends])                                    # [world][block]: the dense halves for r
        cp = parts[:, :Qs * M * 10].view(torch.float32).view(world, Qs, M, 10)
        mp = parts[:, Qs * M * 10:Qs * M * 13].view(world, Qs, M, 3)
        ref_cos, ref_meta = engs[r].rerank_combine(cp, mp, hi - lo)
        assert torch.equal(meta, ref_meta)
        assert torch.equal(cos.view(torch.int32), ref_cos.view(torch.int32))
    for e in engs:
        e.close()


def test_multi_group_launches_on_the_f16_image_equal_the_f32_rows(mods):
    """Engines for >= 512 queries per call serve launches of several 256-query groups from an f16 image of the rows -- the
    values the f32 rows' pass converts in registers -- so that not every group converts them again.  Same products, same
    thresholds, same candidates: the results of 300 / 600 / 1000 queries (2, 3 and 4 groups; launches of 2 + 1 and 4 groups) must
    equal, bit for bit, those of an engine that declined the image (MSR_CFG_NO_ROW_COPY: f32 rows, converted in the kernel),
    and the oracle's for queries of every group.  Rows are not unit norm; chunk-less documents, an exact hit, a zero query,
    a last tile with a partly filled fragment."""
    rng = np.random.default_rng(99)
    n_docs = 50000
    n = rng.integers(0, 8, size=n_docs)
    n[-1] = 3
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    C = int(doc_off[-1])
    emb_t = torch.randn((C, 768), generator=torch.Generator().manual_seed(13))
    emb_t /= emb_t.norm(dim=1, keepdim=True)
    emb_t *= torch.empty(C, 1).uniform_(0.6, 1.8, generator=torch.Generator().manual_seed(14))   # (norms in [0.5, 2]: the default path)
    emb = emb_t.numpy()
    ix = mods["CorpusIndex"](doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                             chunk_ids=np.arange(C, dtype=np.int64), emb=emb, total_docs=n_docs)
    img = mods["DeviceEngine"](ix, max_queries=1024, max_k=100, rerank_max_docs=0)
    raw = mods["DeviceEngine"](ix, max_queries=1024, max_k=100, rerank_max_docs=0, row_copy=False)
    assert img.owned_bytes() - raw.owned_bytes() >= (C + 512) * 768 * 2          # the image (and the fragment-order copy)
    assert img.row_copy_state() == "built" and raw.row_copy_state() == "declined"
    assert img.row_image_state() == "built" and raw.row_image_state() == "declined"
    q = rng.standard_normal((1000, 768)).astype(np.float32) * rng.uniform(0.5, 12, size=(1000, 1)).astype(np.float32)
    q[0] = emb[4567] * 3.0
    q[1] = 0.0
    q[2:60] = emb[rng.integers(0, C, 58)] + 0.4 * q[2:60] / np.linalg.norm(q[2:60], axis=1, keepdims=True)
    q[999] = emb[C - 1] * 0.5                                  # an exact hit in the last tile
    for Q, k in ((300, 100), (600, 10), (1000, 100)):
        a = [x.cpu().numpy() for x in img.dense_topk(q[:Q], k=k)]
        assert img.dense_path() == 256
        b = [x.cpu().numpy() for x in raw.dense_topk(q[:Q], k=k)]
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (Q, k)
        for i in sorted({0, 2, 59, 255, 256, Q // 2, Q - 1}):
            oi, os_, oa = mods["dense_ref"].quick_search(emb, doc_off, q[i], k)
            assert a[3][i] == len(oi)
            np.testing.assert_allclose(a[1][i], os_, rtol=0, atol=1e-5)
            ok = (a[0][i] == oi) | (np.abs(np.r_[np.diff(os_), 1.0]) <= 4e-6) | (np.abs(np.r_[1.0, np.diff(os_)]) <= 4e-6)
            assert ok.all(), (Q, i)
    img.close(); raw.close()
