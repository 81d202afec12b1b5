"""Property tests (hypothesis): random small corpora and queries, HIP path == oracle.  Runs on the MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402

SET = settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


def _mods():
    from msretr.engine import DeviceEngine
    from msretr.index import CorpusIndex
    from oracle import bm25_ref, dense_ref
    return DeviceEngine, CorpusIndex, bm25_ref, dense_ref


@SET
@given(seed=st.integers(0, 10 ** 6), n_docs=st.integers(1, 9000), n_terms=st.integers(1, 60),
       k=st.sampled_from([1, 3, 10, 100, 1000]), min_score=st.sampled_from([0.0, -2.0, 0.7]),
       k1=st.sampled_from([1.2, 0.0, 2.0]), b=st.sampled_from([0.75, 0.0, 1.0]))
def test_bm25_random_corpora(seed, n_docs, n_terms, k, min_score, k1, b):
    DeviceEngine, CorpusIndex, bm25_ref, _ = _mods()
    rng = np.random.default_rng(seed)
    doc_len = rng.integers(1, 300, size=n_docs).astype(np.int32)
    lists, df = [], []
    for t in range(n_terms):
        frac = rng.choice([0.0005, 0.01, 0.2, 0.9])
        d = np.nonzero(rng.random(n_docs) < frac)[0].astype(np.int32)
        lists.append(d); df.append(len(d))
    term_off = np.concatenate([[0], np.cumsum(df)]).astype(np.int64)
    post_doc = np.concatenate(lists) if term_off[-1] else np.zeros(0, np.int32)
    post_tf = rng.integers(1, 6, size=len(post_doc)).astype(np.int32)
    N = n_docs
    idf = np.array([np.float32(np.log10((N - d + 0.5) / (d + 0.5))) for d in df], np.float32)
    ix = dict(doc_ids=np.arange(N, dtype=np.int64) * 2, doc_len=doc_len, term_off=term_off, post_doc=post_doc,
              post_tf=post_tf, idf=idf, avgdl=float(np.float32(doc_len.mean())))
    if len(post_doc) == 0:
        return
    eng = DeviceEngine(CorpusIndex(**ix, total_docs=N, k1=k1, b=b), max_queries=4, max_k=1000)
    qs = [[int(x) for x in rng.integers(-1, n_terms + 1, size=int(rng.integers(0, 7)))] for _ in range(6)]
    doc, score, n = [x.cpu().numpy() for x in eng.bm25_topk(qs, k=k, min_score=min_score)]
    for i, q in enumerate(qs):
        oi, os_ = bm25_ref.topk(ix, q, k, min_score, k1, b)
        assert n[i] == len(oi) and doc[i, :n[i]].tolist() == oi.tolist() and score[i, :n[i]].tolist() == os_.tolist()
    eng.close()


@SET
@given(seed=st.integers(0, 10 ** 6), n_docs=st.integers(1, 1500), max_ch=st.integers(1, 40),
       Q=st.sampled_from([1, 2, 16, 19, 32, 35]), k=st.sampled_from([1, 10, 100]), mc=st.sampled_from([0, 1, 10]),
       layout=st.sampled_from([0, 1]), batched=st.booleans())
def test_dense_random_corpora(seed, n_docs, max_ch, Q, k, mc, layout, batched):
    DeviceEngine, CorpusIndex, _, dense_ref = _mods()
    rng = np.random.default_rng(seed)
    n = rng.integers(0, max_ch + 1, size=n_docs)
    if n.sum() == 0:
        n[0] = 1
    doc_off = np.zeros(n_docs + 1, np.int64); doc_off[1:] = np.cumsum(n)
    emb = rng.standard_normal((int(doc_off[-1]), 768)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    emb[rng.integers(0, len(emb))] *= 2.5                      # a row that is not unit norm
    ix = CorpusIndex(doc_ids=np.arange(n_docs, dtype=np.int64), doc_off=doc_off.astype(np.int32),
                     chunk_ids=np.arange(doc_off[-1], dtype=np.int64), emb=emb, total_docs=n_docs)
    batched = batched and layout == 0
    eng = DeviceEngine(ix, max_queries=32, max_k=100, scan_layout=layout)
    if batched:
        eng.enable_bf16()
    q = (rng.standard_normal((Q, 768)) * rng.uniform(0.1, 20)).astype(np.float32)
    fn = eng.dense_topk_batched if batched else eng.dense_topk
    doc, score, chunk, cnt = [x.cpu().numpy() for x in fn(q, k=k, max_chunks_per_doc=mc)]
    for i in range(Q):
        best, arg = dense_ref.doc_scores(emb, doc_off, q[i], mc)
        oi, os_, _ = dense_ref.quick_search(emb, doc_off, q[i], k, mc)
        assert cnt[i] == len(oi)
        np.testing.assert_allclose(score[i, :cnt[i]], os_, rtol=0, atol=1e-5)
        np.testing.assert_allclose(score[i, :cnt[i]], best[doc[i, :cnt[i]]], rtol=0, atol=1e-5)
        assert all(abs(best[d] - os_[-1]) <= 2e-5 for d in set(doc[i, :cnt[i]].tolist()) ^ set(oi.tolist()))
        lo, hi = doc_off[doc[i, :cnt[i]]], doc_off[doc[i, :cnt[i]] + 1]
        assert np.all((chunk[i, :cnt[i]] >= lo) & (chunk[i, :cnt[i]] < hi))
    eng.close()
