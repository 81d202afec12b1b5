"""The CPU oracle against the golden vectors produced from the reference's own function bodies
(tests/golden/make_goldens.py).  No GPU."""
import json
import os

import numpy as np
import pytest

from oracle import bm25_ref, build_ref, dense_ref, rerank_ref

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(G, name), encoding="utf-8") as f:
        return json.load(f)


# ------------------------------------------------------------------ BM25 known answers
@pytest.mark.parametrize("case", _load("bm25_kat.json"), ids=lambda c: c["name"])
def test_bm25_known_answers(case):
    postings = {t: [tuple(p) for p in pl] for t, pl in case["postings"].items()}
    doc_len = {int(d): l for d, l in case["doc_len"].items()}
    ix, vocab = bm25_ref.index_from_tables(postings, doc_len, case["idf_f32"], case["avgdl_f32"])
    urls_db = {int(d): tuple(v) for d, v in case["urls_db"].items()}
    for q in case["queries"]:
        ids = [vocab.get(t, -1) for t in q["terms"]]
        for fn in (bm25_ref.topk, bm25_ref.topk_literal):
            idx, sc = fn(ix, ids, 10 ** 6, q["min_score"], case["k1"], case["b"])
            assert len(idx) == len(set(idx.tolist()))
        got = bm25_ref.search(ix, ids, q["top_k"], q["min_score"], urls_db, case["k1"], case["b"])
        assert got == q["expected"], (case["name"], q["terms"])     # bitwise: float64 == float64


@pytest.mark.parametrize("name", ["bm25_random_a", "bm25_random_b"])
def test_bm25_random_corpus(name):
    ix = dict(np.load(os.path.join(G, name + ".npz")))
    meta = _load(name + ".json")
    missing = set(meta["missing_from_urlsdb"])
    urls_db = {int(d): ("t", "x") for d in ix["doc_ids"] if int(d) not in missing}
    for qi, q in enumerate(meta["queries"]):
        got = bm25_ref.search(ix, q["terms"], q["top_k"], q["min_score"], urls_db)
        assert [r["doc_id"] for r in got] == q["doc_id"]
        assert [r["score"] for r in got] == q["score"]            # exact float64 equality
        if qi % 9 == 0:                                             # literal form agrees too
            a = bm25_ref.topk(ix, q["terms"], q["top_k"], q["min_score"])
            b = bm25_ref.topk_literal(ix, q["terms"], q["top_k"], q["min_score"])
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


# ------------------------------------------------------------------ cosine
def test_cosine_matches_sklearn_float32():
    z = np.load(os.path.join(G, "cosine.npz"))
    assert str(z["expected_dtype"]) == "float32"
    got = rerank_ref.cosine_f32(z["q"], z["E"])
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, z["expected"], rtol=0, atol=1e-6)
    assert got[7] == 0.0                                            # zero row -> 0, not NaN
    assert abs(float(got[123]) - float(z["single_123"])) < 1e-6


def test_oracle_dense_matches_reference_on_unit_rows():
    """cosine_unit.npz (reference get_new_similarity on unit-norm rows, 40 queries): the oracle's per-chunk cosines,
    per-document maxima and ranking restate it."""
    from oracle import dense_ref
    z = np.load(os.path.join(G, "cosine_unit.npz"))
    E, qs, exp, doc_off = z["E"], z["q"], z["expected"], z["doc_off"].astype(np.int64)
    for i in (0, 7, 39):
        got = rerank_ref.cosine_f32(qs[i], E)
        np.testing.assert_allclose(got, exp[i], rtol=0, atol=1e-6)
        best, arg = dense_ref.doc_scores(E, doc_off, qs[i], 0)
        np.testing.assert_allclose(best, np.maximum.reduceat(exp[i], doc_off[:-1]), rtol=0, atol=1e-6)
        oi, os_, oa = dense_ref.quick_search(E, doc_off, qs[i], 10)
        assert np.all(np.diff(os_) <= 0) and len(oi) == 10


# ------------------------------------------------------------------ rerank chain
def _case_tables(c):
    z = np.load(os.path.join(G, f"rerank_{c['case']}.npz"))
    urls = {int(u[0]): (u[1], u[2], u[3]) for u in c["urls"]}
    order = np.lexsort((z["chunk_id"], z["chunk_doc"]))
    return urls, z["chunk_id"][order], z["chunk_doc"][order], z["emb"][order], z["q"]


def _assert_ranked_equal(got, exp, tol):
    """Same documents in the same order, except inside groups of (near-)equal scores."""
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


@pytest.mark.parametrize("c", _load("rerank_chain.json")["cases"], ids=lambda c: f"case{c['case']}")
def test_rerank_chain(c):
    urls, cid, cdoc, emb, q = _case_tables(c)
    st = {}
    names = ["cos", "cos_norm", "bm25_norm", "blend", None, "positional"]
    chunk_stages = [s for s in c["stages"] if s["target"] != "reranked_documents"]
    assert len(chunk_stages) == len(names)
    for div in (False, True):
        resp, stages = rerank_ref.rerank(urls, cid, cdoc, emb, q, c["doc_ids"], c["similarities"],
                                         diversification=div, return_stages=True)
        exp = c["response"]["div" if div else "nodiv"]
        assert resp["total_documents"] == exp["total_documents"] and resp["total_windows"] == exp["total_windows"]
        _assert_ranked_equal([(d["doc_id"], d["similarity_score"]) for d in resp["document_scores"]],
                             [(d["doc_id"], d["similarity_score"]) for d in exp["document_scores"]], 2e-6)
        gmap = {d["doc_id"]: d for d in resp["document_scores"]}
        for e in exp["document_scores"]:
            g = gmap[e["doc_id"]]
            assert (g["title"], g["url"], g["window_index"]) == (e["title"], e["url"], e["window_index"])
            assert abs(g["original_similarity"] - e["original_similarity"]) < 1e-12
        wmap = {w["doc_id"]: w for w in resp["top_windows"]}
        for w in exp["top_windows"]:
            assert abs(wmap[w["doc_id"]]["similarity_score"] - w["similarity_score"]) < 2e-6
            assert wmap[w["doc_id"]]["window_index"] == w["window_index"]
    # chunk-level stages: reference rows are ordered (doc_id, chunk_id) before AND after the groupby
    rows = stages["rows"]
    for name, s in zip(names, chunk_stages):
        if name is None:
            continue
        assert list(zip(s["doc_id"], s["chunk_id"])) == rows
        col = "old_similarity" if name == "bm25_norm" else "new_similarity"
        np.testing.assert_allclose(stages[name], s[col], rtol=0, atol=2e-6)
    pooled = c["stages"][-1]
    _assert_ranked_equal([(p[0], p[2]) for p in stages["pooled"]],
                         list(zip(pooled["doc_id"], pooled["new_similarity"])), 2e-6)


def test_rerank_empty_is_401():
    assert _load("rerank_chain.json")["empty_error"] == {"type": "HTTPException", "status_code": 401}
    with pytest.raises(LookupError):
        rerank_ref.rerank({}, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 768), np.float32),
                          np.ones(768, np.float32), ["1", "2"], [1.0, 0.5])


# ------------------------------------------------------------------ diversification / windows
def test_diversification():
    d = _load("diversification.json")
    for c in d["cases"]:
        docs = [{"doc_id": str(i), "url": u, "similarity_score": s} for i, u, s in c["input"]]
        got = rerank_ref.hybrid_diversification(docs, top_k=c["top_k"])
        assert [[int(x["doc_id"]), x["similarity_score"]] for x in got] == c["expected"]
    for c in d["domain_cap"]:
        docs = [{"doc_id": i, "url": u, "similarity_score": s} for i, u, s in c["input"]]
        kept, dropped = rerank_ref.apply_domain_cap(docs, 2)
        assert [x["doc_id"] for x in kept] == c["kept"] and [x["doc_id"] for x in dropped] == c["dropped"]
    for u, dom in d["extract_domain"]:
        assert rerank_ref.extract_domain(u) == dom


def test_sliding_windows():
    for c in _load("windows.json"):
        wins = rerank_ref.create_sliding_windows(list(range(c["n"])), c["window"], c["step"])
        assert [w[0] if w else -1 for w in wins] == c["starts"]
        assert [len(w) for w in wins] == c["lens"]


# ------------------------------------------------------------------ dense full scan (self-consistency)
def test_dense_quick_search_matches_bruteforce():
    rng = np.random.default_rng(3)
    n = rng.integers(0, 7, size=200)
    doc_off = np.zeros(201, np.int64); doc_off[1:] = np.cumsum(n)
    emb = rng.standard_normal((int(doc_off[-1]), 768)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    q = rng.standard_normal(768).astype(np.float32) * 4
    for mc in (0, 3):
        idx, sc, arg = dense_ref.quick_search(emb, doc_off, q, 20, mc)
        cos = rerank_ref.cosine_f32(q, emb)
        brute = []
        for d in range(200):
            lo, hi = doc_off[d], doc_off[d + 1]
            if mc:
                hi = min(hi, lo + mc)
            if hi > lo:
                j = lo + int(np.argmax(cos[lo:hi]))
                brute.append((-float(cos[j]), d, j))
        brute.sort()
        assert idx.tolist() == [b[1] for b in brute[:20]]
        assert arg.tolist() == [b[2] for b in brute[:20]]
        assert np.all(np.diff(sc) <= 0)


def test_pandas_shaped_chain_equals_the_restatement():
    """rerank_chain_pandas (the reference's DataFrame / iterrows / groupby shape, used as the literal CPU baseline) gives
    the pooled stage of rerank(), which is pinned to the reference's own stage outputs above."""
    for c in _load("rerank_chain.json")["cases"]:
        urls, cid, cdoc, emb, q = _case_tables(c)
        resp, st = rerank_ref.rerank(urls, cid, cdoc, emb, q, c["doc_ids"], c["similarities"], return_stages=True,
                                     diversification=False, top_k=10 ** 6)
        kept, rows = rerank_ref.fetch_candidates(urls, cid, cdoc, [int(d) for d in c["doc_ids"]])
        old_of = {}
        for d, s_ in zip(c["doc_ids"], c["similarities"]):
            old_of.setdefault(int(d), float(s_))
        kept = [d for d in kept if d in old_of]
        chunk_rows = [(d, int(cid[r]), emb[r]) for d in kept for r in rows[d]]
        lit = rerank_ref.rerank_chain_pandas(chunk_rows, q, [int(d) for d in c["doc_ids"]], c["similarities"])
        pooled = st["pooled"]
        assert [x[0] for x in lit] == [x[0] for x in pooled] and [x[1] for x in lit] == [x[1] for x in pooled]
        assert max(abs(a[2] - b[2]) for a, b in zip(lit, pooled)) < 1e-12


def test_c_port_equals_the_numpy_restatement():
    """oracle/orc_*.c (the C port timed as cpu_baseline) against the numpy restatement that the goldens pin: BM25 bitwise --
    on a corpus large enough for the OpenMP path (one document range per thread), with every thread count -- and the
    dense top-k within float32 rounding."""
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    from oracle import c_oracle, dense_ref
    ix = synthetic_corpus(70000, n_chunks=0, n_terms=30000, seed=31)
    terms, _ = synthetic_queries(ix, 6, seed=32)
    z = {k: np.ascontiguousarray(getattr(ix, k).numpy()) for k in ("term_off", "post_doc", "post_tf", "doc_len", "idf")}
    z["avgdl"] = ix.avgdl
    for threads in (1, 3, 8):
        c_oracle.set_threads(threads)
        for t in terms:
            ut, qtf = bm25_ref.prepare_query(t, z["term_off"])
            a, b = c_oracle.bm25_topk(z, ut, qtf, 1000, 0.0, ix.k1, ix.b)
            oi, os_ = bm25_ref.topk(z, t, 1000)
            assert a.tolist() == oi.tolist() and b.tolist() == os_.tolist()
    small = synthetic_corpus(3000, n_chunks=14000, n_terms=500, seed=33)
    _, qv = synthetic_queries(small, 3, seed=34)
    emb, off = small.emb.numpy(), small.doc_off.numpy()
    for i in range(3):
        a, b, c_ = c_oracle.dense_topk(emb, off, qv[i].numpy(), 50)
        oi, os_, oc = dense_ref.quick_search(emb, off, qv[i].numpy(), 50)
        assert np.abs(b - os_).max() <= 2e-6 and (a == oi).mean() > 0.95


# ------------------------------------------------------------------------------------------------ index build
def _golden_tables(case):
    """The reference's tables of one fixture case as order-free maps."""
    lens = {d: l for d, l in case["doc_stats"]}
    tf = {(d, t): f for d, t, f in case["term_freq"]}
    df = {t: u[0] for t, u in case["term_updates"].items()}
    total = {t: u[1] for t, u in case["term_updates"].items()}
    return lens, tf, df, total


@pytest.mark.parametrize("case", _load("bm25_build.json"), ids=lambda c: c["name"])
def test_index_build_restatement_equals_the_reference_tables(case):
    """oracle/build_ref.process_document_batch against the output of the reference's own BM25._process_document_batch
    (tests/golden/make_goldens.py::bm25_build_fixture): the three tables, row for row and in the reference's row order."""
    docs = [tuple(d) for d in case["documents"]]
    stats, tf, upd = build_ref.process_document_batch(docs, lambda text: text.split())
    assert [list(x) for x in stats] == case["doc_stats"]
    assert [list(x) for x in tf] == case["term_freq"]
    assert {t: list(u) for t, u in upd.items()} == case["term_updates"] and list(upd) == list(case["term_updates"])
    # and laid out for the engine: ascending doc ids, postings ascending inside a term, df = new_docs, sum tf = freq_increase
    z = build_ref.index_from_batches(stats, tf)
    lens, tfm, df, total = _golden_tables(case)
    assert z["doc_ids"].tolist() == sorted(lens) and z["doc_len"].tolist() == [lens[d] for d in sorted(lens)]
    assert set(z["vocab"]) == set(df)
    for t, i in z["vocab"].items():
        lo, hi = int(z["term_off"][i]), int(z["term_off"][i + 1])
        assert hi - lo == df[t] and int(z["post_tf"][lo:hi].sum()) == total[t]
        assert np.all(np.diff(z["post_doc"][lo:hi]) > 0)
        assert {(int(z["doc_ids"][d]), t): int(f) for d, f in zip(z["post_doc"][lo:hi], z["post_tf"][lo:hi])} == \
            {k: v for k, v in tfm.items() if k[1] == t}
    assert z["avgdl"] == float(np.float32(np.mean([l for l in lens.values()])))


@pytest.mark.parametrize("case", _load("bm25_build.json"), ids=lambda c: c["name"])
def test_product_host_builder_equals_the_reference_tables(case):
    """The product's host-side builders (index_build.py: the dict builder and the sort-based one on the CPU device) produce
    the reference's tables for the fixture crawls: same rows as BM25._process_document_batch returned, idf / avgdl as the
    restatement stores them, bit for bit."""
    from msretr.index_build import bm25_index_from_token_ids, bm25_index_from_tokens, normalise_document_text
    docs = [tuple(d) for d in case["documents"]]
    toks = [normalise_document_text(t, x).split() for _, t, x in docs]
    ix = bm25_index_from_tokens([d for d, _, _ in docs], toks)
    stats, tf, _ = build_ref.process_document_batch(docs, lambda text: text.split())
    z = build_ref.index_from_batches(stats, tf, vocab=ix.vocab)
    for name in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf"):
        assert np.array_equal(np.asarray(getattr(ix, name)), z[name]), name
    assert np.array_equal(np.asarray(ix.idf).view(np.uint32), z["idf"].view(np.uint32)) and ix.avgdl == z["avgdl"]
    assert ix.total_docs == z["total_docs"] == len(case["doc_stats"])
    lens, tfm, df, total = _golden_tables(case)
    got = {(int(ix.doc_ids[d]), t): int(f) for t, i in ix.vocab.items()
           for d, f in zip(ix.post_doc[ix.term_off[i]:ix.term_off[i + 1]], ix.post_tf[ix.term_off[i]:ix.term_off[i + 1]])}
    assert got == tfm and {int(d): int(l) for d, l in zip(ix.doc_ids, ix.doc_len)} == lens
    # the sort-based builder from token ids (the form msr_build_postings implements on the GPU), CPU device
    tok_off = np.zeros(len(docs) + 1, np.int64); tok_off[1:] = np.cumsum([len(t) for t in toks])
    tok_ids = np.array([ix.vocab[w] for t in toks for w in t], np.int32)
    sx = bm25_index_from_token_ids([d for d, _, _ in docs], tok_off, tok_ids, len(ix.vocab), device="cpu")
    for name in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf"):
        assert np.array_equal(np.asarray(getattr(sx, name)), z[name]), name
    assert np.array_equal(np.asarray(sx.idf).view(np.uint32), z["idf"].view(np.uint32)) and sx.avgdl == z["avgdl"]


def test_vectorised_idf_equals_the_scalar_formula():
    """index_build evaluates idf for the whole vocabulary at once (numpy float64 log10 -> float32); the restatement's
    scalar math.log10 form must give the same bits for every document frequency."""
    from msretr.index_build import idf_real
    for N in (1, 2, 7, 4999, 1_000_000, 16_777_217):
        top = int(np.float32(N))                      # (N round-trips through a REAL column; df <= N)
        df = np.unique(np.concatenate([np.arange(0, min(top, 3000) + 1), np.linspace(0, top, 2000).astype(np.int64)]))
        got = idf_real(N, df)
        exp = np.array([build_ref.idf_real(N, int(c)) for c in df], np.float32)
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), exp.view(np.uint32)), N
