"""/rerank facade: same request / response fields and error behaviour as the reference endpoint
(reranker/reranker_api.py:141-168, 336-417), with the arithmetic done by msr_rerank on the GPU.

    Reranker(index_or_engine).rerank(doc_ids=[str], similarities=[float], query_embedding=vec)
        -> {"document_scores": [...], "top_windows": [...], "total_documents": int, "total_windows": int}

Host work that stays on the host (it is string / list logic on <= 1000 items in the reference too):
mapping ids, dropping documents whose title/url/text is NULL (the reference's pydantic models reject them,
:376-397) and the domain diversification (:170-236).
"""
from typing import List, Optional, Sequence

import numpy as np

from .engine import RERANK_DEFAULTS, DeviceEngine
from .index import CorpusIndex
from .text import extract_domain

SIMILARITY_DEFAULTS = dict(batch_size=32, smoothing=0.15, diversification=True, top_k=100)   # reranker/config.yaml:25-30


class RerankNotFound(LookupError):
    """No chunk row for the requested ids: HTTP 401 in the reference (reranker_api.py:348-349)."""
    status_code = 401


def _cap_per_domain(docs, limit):
    seen, kept, dropped = {}, [], []
    for d in docs:
        dom = extract_domain(d["url"])
        c = seen.get(dom, 0)
        if c < limit:
            seen[dom] = c + 1
            kept.append(d)
        else:
            dropped.append(d)
    return kept, dropped


def diversify(results, relevance_threshold=0.8, top_k=100):
    """One result per domain among the high-relevance documents (score >= threshold, plus everything from
    their domains), then one per remaining domain; if that leaves fewer than top_k, refill from the dropped
    ones with their scores shifted below the last kept score (reranker_api.py:196-236)."""
    score = lambda d: d["similarity_score"]
    dom = [extract_domain(d["url"]) for d in results]
    hi_dom = {m for d, m in zip(results, dom) if score(d) >= relevance_threshold}
    high = sorted((d for d, m in zip(results, dom) if score(d) >= relevance_threshold or m in hi_dom),
                  key=score, reverse=True)
    med = sorted((d for d, m in zip(results, dom) if score(d) < relevance_threshold and m not in hi_dom),
                 key=score, reverse=True)
    keep_hi, drop_hi = _cap_per_domain(high, 1)
    keep_med, drop_med = _cap_per_domain(med, 1)
    final = sorted(keep_hi + keep_med[: top_k - len(keep_hi)], key=score, reverse=True)
    if len(final) < top_k:
        refill = sorted(drop_hi + drop_med, key=score, reverse=True)[: top_k - len(final)]
        if refill:
            delta = score(refill[0]) - score(final[-1]) + 1e-4
            for d in refill:
                d["similarity_score"] = max(0.0, d["similarity_score"] - delta)
            final.extend(refill)
    return sorted(final, key=score, reverse=True)


class Reranker:
    def __init__(self, source, config: Optional[dict] = None, encoder=None, device=0, **engine_kw):
        self.engine = source if isinstance(source, DeviceEngine) else DeviceEngine(source, device=device, **engine_kw)
        self.index: CorpusIndex = self.engine.index
        self.cfg = dict(SIMILARITY_DEFAULTS)
        self.cfg.update(config or {})
        self.encoder = encoder
        ids = self.index.doc_ids
        ids = ids.cpu().numpy() if hasattr(ids, "cpu") else np.asarray(ids)
        self._pos = {int(d): i for i, d in enumerate(ids)}
        self._ids = ids
        cid = self.index.chunk_ids
        self._chunk_ids = cid.cpu().numpy() if hasattr(cid, "cpu") else np.asarray(cid)

    def _embed(self, query, query_embedding):
        if query_embedding is not None:
            return np.asarray(query_embedding, np.float32)
        if self.encoder is None:
            raise ValueError("no query_embedding given and no encoder configured (the reference loads a "
                             "sentence-transformers model by name, reranker_api.py:137-139)")
        enc = self.encoder.encode if hasattr(self.encoder, "encode") else self.encoder
        return np.asarray(enc(query), np.float32)

    def rerank_batch(self, requests: Sequence[dict]):
        """requests: dicts with doc_ids, similarities, query / query_embedding.  One GPU call for all."""
        M = self.engine.rerank_max_docs
        Q = len(requests)
        cand = np.full((Q, M), -1, np.int32)
        bm = np.zeros((Q, M), np.float64)
        n = np.zeros(Q, np.int32)
        qv = np.zeros((Q, 768), np.float32)
        for r, req in enumerate(requests):
            if req.get("similarities") is None:
                raise ValueError("similarities are required (the reference fails with HTTP 500 without them)")
            if len(req["doc_ids"]) != len(req["similarities"]):
                raise ValueError("doc_ids and similarities differ in length")
            seen = set()
            j = 0
            for d, s in zip(req["doc_ids"], req["similarities"]):
                d = int(d)
                if d in seen or d not in self._pos:
                    continue
                seen.add(d)
                if j >= M:
                    raise ValueError(f"more than {M} candidates; raise rerank_max_docs")
                cand[r, j], bm[r, j] = self._pos[d], float(s)
                j += 1
            n[r] = j
            qv[r] = self._embed(req.get("query"), req.get("query_embedding"))
        p = dict(RERANK_DEFAULTS)
        p["smoothing"] = self.cfg["smoothing"]
        out = [x.cpu().numpy() for x in self.engine.rerank(qv, cand, bm, n, **p)]
        return [self._response(r, out) for r in range(Q)]

    def rerank(self, doc_ids: List[str], similarities: Optional[List[float]] = None, query: Optional[str] = None,
               query_embedding=None) -> dict:
        return self.rerank_batch([dict(doc_ids=doc_ids, similarities=similarities, query=query,
                                       query_embedding=query_embedding)])[0]

    def _response(self, r, out):
        doc, score, orig, chunk, cnt, rows = out
        if cnt[r] == 0:
            raise RerankNotFound("No documents found for the provided doc_ids")
        ix = self.index
        scored = []
        for j in range(int(cnt[r])):
            i = int(doc[r, j])
            title = ix.titles[i] if ix.titles is not None else ""
            url = ix.urls[i] if ix.urls is not None else ""
            text = ix.texts[i] if ix.texts is not None else ""
            if title is None or url is None or text is None:
                continue                                       # pydantic would reject the NULL (:376-397)
            did = str(int(self._ids[i]))
            s = float(score[r, j])
            win = {"text": text, "similarity_score": s, "doc_id": did, "title": title,
                   "window_index": int(self._chunk_ids[int(chunk[r, j])])}
            scored.append({"doc_id": did, "title": title, "url": url, "similarity_score": s,
                           "original_similarity": float(orig[r, j]), "most_relevant_window": win})
        top_k = int(self.cfg["top_k"])
        ranked = diversify(scored, top_k=top_k) if self.cfg.get("diversification", False) else scored[:top_k]
        return {"document_scores": ranked, "top_windows": [d["most_relevant_window"] for d in ranked[:top_k]],
                "total_documents": int(rows[r]), "total_windows": top_k}
