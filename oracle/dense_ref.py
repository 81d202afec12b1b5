"""CPU restatement of the dense full-scan retriever.  TEST INFRASTRUCTURE ONLY.

`Retriever.quick_search(query, top_k, return_unique_docs=True)` is called at
/root/reference/search_api.py:60,87 but retriever.py is absent from the reference snapshot (SURVEY.md
F2/F3).  Its semantics are taken from what the rest of the reference says about it:
  * brute-force similarity of the query against every chunk embedding ("Sequential Search",
    indexer/README.md:186-190; inner-product index `ip_idx`, indexer/indexer.py:66-67)
  * sim(d, q) = max_i cos(q, c_i) over the document's chunks (Project_Report.pdf p.2;
    reranker_api.py:370 for the per-document arg-max)
  * cosine as in reranker_api.py:285 (see rerank_ref.cosine_f32)
Selection rule (this build's definition, "parity unpinned"): score descending, ties by ascending doc
index, documents without chunks are never returned.
"""
import numpy as np

from .rerank_ref import cosine_f32


def doc_scores(emb, doc_off, qvec, max_chunks=0, block=262144):
    """-> (best[N] float32 with -inf for chunk-less docs, best_chunk[N] int64 row index or -1)."""
    doc_off = np.asarray(doc_off, np.int64)
    N = len(doc_off) - 1
    C = int(doc_off[-1])
    cos = np.empty(C, np.float32)
    for s in range(0, C, block):
        cos[s:s + block] = cosine_f32(qvec, emb[s:s + block])
    best = np.full(N, -np.inf, np.float32)
    arg = np.full(N, -1, np.int64)
    n = np.diff(doc_off)
    if max_chunks > 0:
        pos = np.arange(C) - np.repeat(doc_off[:-1], n)
        cos = np.where(pos < max_chunks, cos, -np.inf).astype(np.float32)
    nz = np.nonzero(n > 0)[0]
    if len(nz):
        best[nz] = np.maximum.reduceat(cos, doc_off[nz])
        # first maximum inside each segment
        seg = np.repeat(np.arange(N), n)
        is_max = cos == best[seg]
        first = np.full(N, C, np.int64)
        np.minimum.at(first, seg[is_max], np.nonzero(is_max)[0])
        arg[nz] = first[nz]
    return best, arg


def quick_search(emb, doc_off, qvec, top_k=100, max_chunks=0):
    """-> (doc index[<=k], score[<=k] float32, best chunk row[<=k])."""
    best, arg = doc_scores(emb, doc_off, qvec, max_chunks)
    cand = np.nonzero(np.isfinite(best))[0]
    order = np.argsort(-best[cand], kind="stable")[:top_k]
    sel = cand[order]
    return sel.astype(np.int64), best[sel], arg[sel]


def merge_topk(parts, top_k):
    """Deterministic merge of per-shard (global doc index, score) lists: score desc, index asc."""
    idx = np.concatenate([p[0] for p in parts])
    sc = np.concatenate([p[1] for p in parts])
    order = np.lexsort((idx, -sc.astype(np.float64)))[:top_k]
    return idx[order], sc[order]
