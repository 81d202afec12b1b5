"""CPU restatement of the reference's BM25 stage.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/indexer/bm25_indexer.py:
  * query term handling (qtf over all tokens, unique terms in first-occurrence order, unknown terms
    skipped, no valid term -> [])                                             :395-432
  * tf component and score accumulation in float64, query-term order           :458-478
  * `score >= min_score` filter applied only to documents touched by a posting  :461,480
  * stable descending sort over candidates that arrive in ascending doc_id      :445,484-485
  * urlsDB join after the top_k cut, snippet format                             :490-512
Index arrays use the layout of the engine (CSR postings over a dense document index that is the rank of
doc_id in ascending order), so "ascending dense index" == "ascending doc_id".
"""
from collections import defaultdict

import numpy as np


def prepare_query(term_ids, term_off):
    """term_ids: list of int (term id, or any id outside [0, V) / with df == 0 for an unknown term).
    Returns (unique valid term ids in first-occurrence order, their query frequencies).
    bm25_indexer.py:405-431."""
    V = len(term_off) - 1
    qtf = defaultdict(int)
    for t in term_ids:
        qtf[int(t)] += 1
    uniq = list(qtf.keys())
    valid = [t for t in uniq if 0 <= t < V and term_off[t + 1] > term_off[t]]
    return valid, [qtf[t] for t in valid]


def scores_dense(ix, uterms, qtf, k1=1.2, b=0.75):
    """Vectorised float64 TAAT with the reference's operation order (bm25_indexer.py:472-478).
    Returns (acc[N] float64, touched[N] bool)."""
    N = len(ix["doc_len"])
    acc = np.zeros(N, np.float64)
    touched = np.zeros(N, bool)
    avgdl = float(np.float32(ix["avgdl"]))               # REAL column -> python float
    dl_all = ix["doc_len"].astype(np.float64)
    for t, f in zip(uterms, qtf):
        lo, hi = int(ix["term_off"][t]), int(ix["term_off"][t + 1])
        d = ix["post_doc"][lo:hi]
        tf = ix["post_tf"][lo:hi].astype(np.float64)
        idf = float(np.float32(ix["idf"][t]))
        dl = dl_all[d]
        # tf_component = (tf * (k1 + 1)) / (tf + k1 * (1 - b + b * doc_length / avg_doc_length))
        comp = (tf * (k1 + 1)) / (tf + k1 * ((1 - b) + (b * dl) / avgdl))
        acc[d] = acc[d] + (idf * comp) * float(f)        # a document occurs once per posting list
        touched[d] = True
    return acc, touched


def topk(ix, term_ids, top_k, min_score=0.0, k1=1.2, b=0.75):
    """-> (dense doc index[<=top_k] int64, score[<=top_k] float64); score desc, ties by index asc."""
    uterms, qtf = prepare_query(term_ids, ix["term_off"])
    if not uterms:
        return np.zeros(0, np.int64), np.zeros(0, np.float64)
    acc, touched = scores_dense(ix, uterms, qtf, k1, b)
    cand = np.nonzero(touched & (acc >= min_score))[0]
    order = np.argsort(-acc[cand], kind="stable")        # stable: equal scores keep ascending index
    sel = cand[order[:top_k]]
    return sel.astype(np.int64), acc[sel]


def topk_literal(ix, term_ids, top_k, min_score=0.0, k1=1.2, b=0.75):
    """Reference-shaped pure-Python loops (dict of dicts, per-document inner loop over the valid terms).
    Small inputs only; used to cross-check the vectorised form and as the 'literal' CPU baseline."""
    uterms, qtf = prepare_query(term_ids, ix["term_off"])
    if not uterms:
        return np.zeros(0, np.int64), np.zeros(0, np.float64)
    qtf = dict(zip(uterms, qtf))
    avgdl = float(np.float32(ix["avgdl"]))
    rows = []
    for t in uterms:
        lo, hi = int(ix["term_off"][t]), int(ix["term_off"][t + 1])
        for d, tf in zip(ix["post_doc"][lo:hi].tolist(), ix["post_tf"][lo:hi].tolist()):
            rows.append((d, t, tf, int(ix["doc_len"][d])))
    rows.sort(key=lambda r: r[0])                        # ORDER BY tf.doc_id
    doc_terms, doc_lengths = defaultdict(dict), {}
    for d, t, tf, dl in rows:
        doc_terms[d][t] = tf
        doc_lengths[d] = dl
    out = []
    for d, tfs in doc_terms.items():
        dl = doc_lengths[d]
        s = 0.0
        for t in uterms:
            if t in tfs:
                tf = tfs[t]
                idf = float(np.float32(ix["idf"][t]))
                comp = (tf * (k1 + 1)) / (tf + k1 * (1 - b + b * dl / avgdl))
                s += idf * comp * qtf[t]
        if s >= min_score:
            out.append((d, s))
    out.sort(key=lambda x: x[1], reverse=True)
    out = out[:top_k]
    return np.array([d for d, _ in out], np.int64), np.array([s for _, s in out], np.float64)


def search(ix, term_ids, top_k=1000, min_score=0.0, urls_db=None, k1=1.2, b=0.75):
    """Full BM25.search result shape: list of {"doc_id", "score", "text_snippet"}.
    urls_db: dict doc_id -> (title|None, text); documents absent from it are dropped after the cut."""
    idx, sc = topk(ix, term_ids, top_k, min_score, k1, b)
    res = []
    for i, s in zip(idx.tolist(), sc.tolist()):
        doc_id = int(ix["doc_ids"][i])
        if urls_db is None:
            res.append({"doc_id": doc_id, "score": s, "text_snippet": None})
            continue
        if doc_id in urls_db:
            title, text = urls_db[doc_id]
            snip = f"{title or 'N/A'}: {text[:200]}"
            if len(text or "") > 200:
                snip += "..."
            res.append({"doc_id": doc_id, "score": s, "text_snippet": snip})
    return res


def index_from_tables(postings, doc_len, idf, avgdl):
    """Build engine-layout arrays from reference-style tables (term -> [(doc_id, tf)], doc_id -> len,
    term -> idf or None).  Returns (ix, vocab) with vocab: term -> id.  NULL idf -> 0.0 (:426)."""
    doc_ids = np.array(sorted(doc_len), np.int64)
    rank = {int(d): i for i, d in enumerate(doc_ids)}
    vocab = {t: i for i, t in enumerate(postings)}
    term_off = np.zeros(len(vocab) + 1, np.int64)
    pd_, ptf = [], []
    for t, i in vocab.items():
        plist = sorted((rank[int(d)], int(tf)) for d, tf in postings[t] if int(d) in rank)
        pd_ += [p[0] for p in plist]
        ptf += [p[1] for p in plist]
        term_off[i + 1] = len(pd_)
    ix = dict(doc_ids=doc_ids, doc_len=np.array([doc_len[int(d)] for d in doc_ids], np.int32),
              term_off=term_off, post_doc=np.array(pd_, np.int32), post_tf=np.array(ptf, np.int32),
              idf=np.array([(idf[t] or 0.0) for t in vocab], np.float32), avgdl=np.float32(avgdl),
              total_docs=np.int64(len(doc_ids)))
    return ix, vocab
