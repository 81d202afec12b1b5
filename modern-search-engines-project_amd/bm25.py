"""BM25 facade with the reference's signature and return shape (indexer/bm25_indexer.py:57,383-514).

    BM25(index_or_engine, k1=1.2, b=0.75).search(query, top_k=1000, min_score=0.0)
        -> [{"doc_id": int, "score": float, "text_snippet": str}, ...]

The reference opens a DuckDB file and re-reads postings per query; here the postings are already in HBM
(DeviceEngine) and the scoring loop + sort + cut run as msr_bm25_topk.  The host keeps what is host work
in the reference too: tokenising the query string and formatting the snippet.
"""
from typing import Callable, List, Optional, Sequence, Union

from .engine import DeviceEngine
from .index import CorpusIndex
from .text import simple_tokenize


class BM25:
    def __init__(self, source: Union[CorpusIndex, DeviceEngine], k1: float = 1.2, b: float = 0.75,
                 tokenizer: Optional[Callable[[str], List[str]]] = None, device=0, **engine_kw):
        if isinstance(source, DeviceEngine):
            self.engine = source
        else:
            source.k1, source.b = k1, b
            self.engine = DeviceEngine(source, device=device, **engine_kw)
        self.index = self.engine.index
        self.k1, self.b = self.index.k1, self.index.b
        self._tokenize = tokenizer or simple_tokenize

    # -- engine-level entry points (usable without spaCy: pre-tokenised terms or term ids) ----------
    def search_terms(self, terms: Sequence[Union[str, int]], top_k: int = 1000, min_score: float = 0.0):
        return self.search_terms_batch([terms], top_k, min_score)[0]

    def search_terms_batch(self, term_lists, top_k: int = 1000, min_score: float = 0.0):
        """-> per query: list of (doc_id, score) in rank order (before the urlsDB join)."""
        ids = [self.index.term_ids(t) for t in term_lists]
        doc, score, n = self.engine.bm25_topk(ids, k=top_k, min_score=min_score)
        doc, score, n = doc.cpu().numpy(), score.cpu().numpy(), n.cpu().numpy()
        doc_ids = self.index.doc_ids
        doc_ids = doc_ids.cpu().numpy() if hasattr(doc_ids, "cpu") else doc_ids
        return [[(int(doc_ids[d]), float(s)) for d, s in zip(doc[q, :n[q]], score[q, :n[q]])] for q in range(len(ids))]

    # -- the reference's method ------------------------------------------------------------------------
    def search(self, query: str, top_k: int = 1000, min_score: float = 0.0):
        query_terms = self._tokenize(query)
        if not query_terms:
            return []                                              # bm25_indexer.py:396-397
        return self._finish(self.search_terms(query_terms, top_k, min_score))

    def _finish(self, ranked):
        """urlsDB join after the cut: documents without a row are dropped, snippet = title + 200 chars
        (bm25_indexer.py:490-512)."""
        ix = self.index
        if ix.urls is None:
            return [{"doc_id": d, "score": s, "text_snippet": None} for d, s in ranked]
        pos = getattr(self, "_pos", None)
        if pos is None:
            ids = ix.doc_ids.cpu().numpy() if hasattr(ix.doc_ids, "cpu") else ix.doc_ids
            pos = self._pos = {int(d): i for i, d in enumerate(ids)}
        out = []
        for d, s in ranked:
            i = pos[d]
            if ix.urls[i] is None:                                 # no urlsDB row
                continue
            title, text = ix.titles[i], ix.texts[i]
            snip = f"{title or 'N/A'}: {text[:200]}"
            if len(text or "") > 200:
                snip += "..."
            out.append({"doc_id": d, "score": s, "text_snippet": snip})
        return out
