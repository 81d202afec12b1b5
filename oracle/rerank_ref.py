"""CPU restatement of the reference's dense rerank / fuse stage.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/reranker/reranker_api.py:
  * candidate fetch: URL-without-query-string dedup keeping MIN(id); first <=10 chunks per document
    (the reference's ROW_NUMBER has no ORDER BY -- this build defines it as ascending chunk_id)   :27-63
  * cosine = sklearn.metrics.pairwise.cosine_similarity in float32 (row-normalise both sides with
    zero norms replaced by 1, then dot); scikit-learn 1.7.x `normalize` + `safe_sparse_dot`       :273-287
  * min-max over ALL chunk rows of the request, in Python floats                                  :289-296
  * blend new*(1-smoothing) + old*smoothing                                                        :362
  * positional weighting of each document's best chunk                                             :299-334
  * per-document arg-max (first maximum), descending sort                                          :370-372
  * domain diversification and response assembly                                                   :170-236, 374-412
Where the reference leaves an order unspecified (pandas quicksort on ties, :372) this restatement orders
ties by ascending doc_id; tests compare tie groups as sets.
"""
from urllib.parse import urlparse

import numpy as np

MAX_CHUNKS_PER_DOC = 10          # reranker_api.py:58
MAX_BOOST, MAX_DECAY = 0.1, 0.05  # reranker_api.py:317-318


def cosine_f32(q, E):
    """cosine_similarity(q[1,D], E[n,D])[0] as scikit-learn computes it for float32 input."""
    q = np.asarray(q, np.float32).reshape(1, -1)
    E = np.asarray(E, np.float32)
    qn = np.sqrt(np.einsum("ij,ij->i", q, q)).astype(np.float32)
    qn[qn == 0.0] = 1.0
    en = np.sqrt(np.einsum("ij,ij->i", E, E)).astype(np.float32)
    en[en == 0.0] = 1.0
    return ((E / en[:, None]) @ (q / qn[:, None])[0]).astype(np.float32)


def normalise(sims):
    """reranker_api.py:289-296 (python floats; all-equal -> zeros)."""
    lo, hi = min(sims), max(sims)
    if hi == lo:
        return [0.0 for _ in sims]
    return [(s - lo) / (hi - lo) for s in sims]


def extract_domain(url):
    try:
        return urlparse(url).netloc.lower()
    except Exception:                                   # reranker_api.py:175-176
        return "defaultdomain"


def apply_domain_cap(results, max_per_domain):
    counts, kept, dropped = {}, [], []
    for doc in results:
        dom = extract_domain(doc["url"])
        if counts.get(dom, 0) < max_per_domain:
            kept.append(doc)
            counts[dom] = counts.get(dom, 0) + 1
        else:
            dropped.append(doc)
    return kept, dropped


def hybrid_diversification(results, relevance_threshold=0.8, top_k=100):
    """reranker_api.py:196-236.  `results`: list of dicts with 'url' and 'similarity_score'
    (sorted descending); dicts of the filled tail are modified in place like the reference's objects."""
    key = lambda d: d["similarity_score"]
    hi_dom = {extract_domain(d["url"]) for d in results if key(d) >= relevance_threshold}
    med_dom = {extract_domain(d["url"]) for d in results if key(d) < relevance_threshold} - hi_dom
    high = [d for d in results if key(d) >= relevance_threshold or extract_domain(d["url"]) in hi_dom]
    med = [d for d in results if key(d) < relevance_threshold and extract_domain(d["url"]) in med_dom]
    high = sorted(high, key=key, reverse=True)
    med = sorted(med, key=key, reverse=True)
    div_high, drop_high = apply_domain_cap(high, 1)
    remaining = top_k - len(div_high)
    div_med, drop_med = apply_domain_cap(med, 1)
    final = sorted(div_high + div_med[:remaining], key=key, reverse=True)
    rest = sorted(drop_high + drop_med, key=key, reverse=True)
    if len(final) < top_k:
        additional = rest[: top_k - len(final)]
        if additional:
            eps = 1e-4
            delta = additional[0]["similarity_score"] - final[-1]["similarity_score"] + eps
            for d in additional:
                d["similarity_score"] = max(0.0, d["similarity_score"] - delta)
            final.extend(additional)
    return sorted(final, key=key, reverse=True)


def fetch_candidates(urls, chunk_id, chunk_doc, doc_ids):
    """-> list of kept doc ids (ascending) and, per kept doc, the row indices of its first <=10 chunks.
    urls: dict id -> (url, title, text).  chunk arrays sorted by (doc, chunk_id)."""
    want = sorted({int(d) for d in doc_ids} & set(urls))
    groups = {}
    for i in want:                                       # ascending id => MIN(id) wins its URL group
        u = urls[i][0]
        groups.setdefault(u[: u.index("?")] if "?" in u else u, i)
    kept = sorted(groups.values())
    rows = {}
    for d in kept:
        lo = int(np.searchsorted(chunk_doc, d, "left"))
        hi = int(np.searchsorted(chunk_doc, d, "right"))
        rows[d] = list(range(lo, min(hi, lo + MAX_CHUNKS_PER_DOC)))
    return [d for d in kept if rows[d]], rows


def positional_adjust(new, n):
    """Apply reranker_api.py:299-334 to one document's chunk scores `new` (chunk_id order)."""
    if n == 1:
        return new
    best = max(range(n), key=lambda i: (new[i], -i))     # idxmax: first maximum
    ratio = best / max(1, n - 1)
    adj = MAX_BOOST - (MAX_BOOST + MAX_DECAY) * ratio
    new = list(new)
    new[best] = max(0.0, min(1.0, new[best] + adj))
    return new


def chain_from_cosines(docs, n_rows, bm25, cos, smoothing=0.15):
    """reranker_api.py:360-372 on the cosines of the candidates' chunk rows: min-max of the cosines and of the BM25 scores
    (repeated per chunk row) over ALL rows, blend, positional weighting per document, per-document first maximum.
    docs: the kept documents in the order of their rows; n_rows[i] chunk rows of docs[i] (>= 1), bm25[i] its stage-1 score;
    cos: one float per chunk row, documents in that order, chunks in chunk_id order.
    -> ([(doc, new_similarity, old_similarity, index of the winning chunk within the document)] in `docs` order, stages)."""
    new = normalise([float(x) for x in cos])
    old = normalise([float(b) for b, n in zip(bm25, n_rows) for _ in range(n)])
    st = {"cos_norm": list(new), "bm25_norm": list(old)}
    new = [a * (1 - smoothing) + o * smoothing for a, o in zip(new, old)]
    st["blend"] = list(new)
    out, pos, p = [], [], 0
    for d, n in zip(docs, n_rows):
        adj = positional_adjust(new[p:p + n], n)
        pos += adj
        best = max(range(n), key=lambda i: (adj[i], -i))
        out.append((d, adj[best], old[p + best], best))
        p += n
    st["positional"] = pos
    return out, st


def rerank(urls, chunk_id, chunk_doc, emb, qvec, doc_ids, similarities, smoothing=0.15, top_k=100,
           diversification=True, return_stages=False):
    """Restatement of the /rerank endpoint body (reranker_api.py:337-412).  Returns the response as a
    dict; raises LookupError (HTTP 401 in the reference, :348-349) when no chunk row is found."""
    kept, rows = fetch_candidates(urls, chunk_id, chunk_doc, doc_ids)
    if not kept:
        raise LookupError("No documents found for the provided doc_ids")
    old_of = {}
    for d, s in zip(doc_ids, similarities):
        old_of.setdefault(int(d), float(s))
    kept = [d for d in kept if d in old_of]
    flat = [(d, r) for d in kept for r in rows[d]]
    cos = cosine_f32(qvec, emb[[r for _, r in flat]])
    pooled, st = chain_from_cosines(kept, [len(rows[d]) for d in kept], [old_of[d] for d in kept], cos, smoothing)
    stages = {"cos": [float(x) for x in cos], "cos_norm": st["cos_norm"], "bm25_norm": st["bm25_norm"], "blend": st["blend"],
              "positional": st["positional"]}
    out = [{"doc_id": str(d), "title": urls[d][1], "url": urls[d][0], "similarity_score": s_, "original_similarity": o_,
            "window_index": int(chunk_id[rows[d][b_]]), "window_score": s_} for d, s_, o_, b_ in pooled]
    stages["rows"] = [(d, int(chunk_id[r])) for d, r in flat]
    out.sort(key=lambda x: (-x["similarity_score"], int(x["doc_id"])))
    stages["pooled"] = [(int(x["doc_id"]), x["window_index"], x["similarity_score"], x["original_similarity"]) for x in out]
    # DocumentScore/WindowScore declare title, url and text as `str`; a NULL in any of them makes the
    # pydantic constructor raise and the reference skips that document (reranker_api.py:376-397).
    out = [x for x in out if None not in urls[int(x["doc_id"])]]
    ranked = hybrid_diversification(out, top_k=top_k) if diversification else out[:top_k]
    resp = {"document_scores": [{k: x[k] for k in ("doc_id", "title", "url", "similarity_score",
                                                    "original_similarity", "window_index")} for x in ranked],
            "top_windows": [{"doc_id": x["doc_id"], "window_index": x["window_index"],
                             "similarity_score": x["window_score"]} for x in ranked[:top_k]],
            "total_documents": len(flat), "total_windows": top_k}
    return (resp, stages) if return_stages else resp


def rerank_chain_pandas(chunk_rows, qvec, doc_ids, similarities, smoothing=0.15, batch_size=32):
    """The /rerank arithmetic in the reference's OWN shape -- a pandas DataFrame of chunk rows, a merge with the BM25
    scores, iterrows() feeding 32-row cosine batches, Python-list min-max, groupby().apply() for the positional weight,
    groupby().idxmax() and sort_values() (reranker_api.py:357-372, with :273-334) -- restated, not imported.  Used as the
    'literal' CPU baseline and cross-checked against rerank() above.
    chunk_rows: list of (doc_id, chunk_id, embedding[768]) for the candidates' first <= 10 chunks.
    -> list of (doc_id, chunk_id, new_similarity, old_similarity) sorted by new_similarity descending."""
    import pandas as pd
    df = pd.DataFrame({"doc_id": [int(r[0]) for r in chunk_rows], "chunk_id": [int(r[1]) for r in chunk_rows],
                       "embedding": [r[2] for r in chunk_rows]})
    sim_df = pd.DataFrame({"doc_id": [int(d) for d in doc_ids], "old_similarity": [float(x) for x in similarities]})
    df = df.merge(sim_df, on="doc_id", how="left")                                  # :357
    sims, batch = [], []
    for _, row in df.iterrows():                                                    # :273-287
        batch.append(np.array(row["embedding"]))
        if len(batch) == batch_size:
            sims.extend(cosine_f32(qvec, np.array(batch)).tolist())
            batch = []
    if batch:
        sims.extend(cosine_f32(qvec, np.array(batch)).tolist())
    df["new_similarity"] = normalise(sims)                                          # :358-360
    df["old_similarity"] = normalise(df["old_similarity"].tolist())                 # :361
    df["new_similarity"] = df["new_similarity"] * (1 - smoothing) + df["old_similarity"] * smoothing   # :362

    def weigh(group):                                                               # :299-334
        if len(group) == 1:
            return group
        group = group.sort_values("chunk_id").reset_index(drop=True)
        best = int(group["new_similarity"].idxmax())
        ratio = best / max(1, len(group) - 1)
        adj = MAX_BOOST - (MAX_BOOST + MAX_DECAY) * ratio
        group.loc[best, "new_similarity"] = float(np.clip(group.loc[best, "new_similarity"] + adj, 0.0, 1.0))
        return group

    df = pd.concat([weigh(g) for _, g in df.groupby("doc_id", sort=True)], ignore_index=True)   # :367 (groupby.apply)
    best = df.loc[df.groupby("doc_id")["new_similarity"].idxmax()]                  # :370
    best = best.sort_values(["new_similarity", "doc_id"], ascending=[False, True])  # :372 (ties: this build's doc order)
    return [(int(r.doc_id), int(r.chunk_id), float(r.new_similarity), float(r.old_similarity)) for r in best.itertuples()]


def create_sliding_windows(tokens, window_size, step_size):
    """reranker_api.py:239-260 (identical copy at indexer/embedder.py:65-87)."""
    if len(tokens) <= window_size:
        return [tokens]
    windows = [tokens[i:i + window_size] for i in range(0, len(tokens) - window_size + 1, step_size)]
    last = len(tokens) - window_size
    if last >= 0 and last % step_size != 0:
        windows.append(tokens[last:last + window_size])
    return windows
