"""Retriever facade: the query -> ranked-documents API of the reference.

  * Retriever(embedder, indexer, db_path).quick_search(query, top_k, return_unique_docs=True)
        the call shape left in search_api.py:60,87 (retriever.py itself is missing from the reference):
        dense retrieval over ALL chunk embeddings, one result per document (max over its chunks).
  * Retriever.search(...) / batch_search(...) / batch_search_to_file(...)
        the live two-stage path of search_api.py:69-152 and :204-367: preprocess_query -> BM25 top-1000 ->
        rerank -> top-100, formatted for the UI / as `qnum<TAB>rank<TAB>url<TAB>score` lines.
"""
import numpy as np

from .bm25 import BM25
from .engine import DeviceEngine
from .index import CorpusIndex
from .reranker import Reranker, RerankNotFound
from .text import extract_domain_topic, format_result_line, preprocess_query, read_queries_file

TOP_K_RETRIEVAL = 1000     # config.py:13
TOP_K_RERANKING = 100      # config.py:14


class Retriever:
    def __init__(self, embedder=None, indexer=None, db_path=None, tokenizer=None, device=0, **engine_kw):
        if isinstance(indexer, DeviceEngine):
            self.engine = indexer
        else:
            index = indexer if isinstance(indexer, CorpusIndex) else CorpusIndex.from_duckdb(db_path)
            self.engine = DeviceEngine(index, device=device, **engine_kw)
        self.index = self.engine.index
        self.embedder = embedder
        self.bm25 = BM25(self.engine, tokenizer=tokenizer) if self.index.term_off is not None else None
        self.reranker = Reranker(self.engine, encoder=embedder) if self.index.doc_off is not None else None
        ids = self.index.doc_ids
        self._ids = ids.cpu().numpy() if hasattr(ids, "cpu") else np.asarray(ids)

    def _embed(self, query, query_embedding=None):
        if query_embedding is not None:
            return np.asarray(query_embedding, np.float32)
        if self.embedder is None:
            raise ValueError("a query string needs an embedder (callable or object with .encode)")
        enc = self.embedder.encode if hasattr(self.embedder, "encode") else self.embedder
        return np.asarray(enc(query), np.float32)

    # ------------------------------------------------------------------ dense full scan
    def quick_search_batch(self, queries=None, top_k=10, return_unique_docs=True, query_embeddings=None,
                           max_chunks_per_doc=0):
        if not return_unique_docs:
            return self._chunk_search_batch(queries, top_k, query_embeddings)
        qv = np.stack([self._embed(q, None if query_embeddings is None else query_embeddings[i])
                       for i, q in enumerate(queries if queries is not None else [None] * len(query_embeddings))])
        doc, score, chunk, n = [x.cpu().numpy() for x in self.engine.dense_topk(qv, k=top_k, max_chunks_per_doc=max_chunks_per_doc)]
        ix = self.index
        cid = ix.chunk_ids.cpu().numpy() if hasattr(ix.chunk_ids, "cpu") else np.asarray(ix.chunk_ids)
        out = []
        for r in range(len(qv)):
            rows = []
            for j in range(int(n[r])):
                i = int(doc[r, j])
                rows.append({"rank": j + 1, "doc_id": int(self._ids[i]), "score": float(score[r, j]),
                             "best_chunk_id": int(cid[int(chunk[r, j])]),
                             "url": ix.urls[i] if ix.urls is not None else None,
                             "title": ix.titles[i] if ix.titles is not None else None})
            out.append(rows)
        return out

    def _chunk_search_batch(self, queries, top_k, query_embeddings):
        """return_unique_docs=False: the top_k CHUNKS by cosine, several per document allowed (the other half of the call
        shape at search_api.py:87; retriever.py itself is absent from the reference, so the row format is ours: the
        unique-document row plus `chunk_id`).  Runs the same scan kernels over a view of the corpus in which every chunk is
        its own document (built once, on first use; the embedding matrix is shared, not copied)."""
        from .engine import DeviceEngine
        from .index import CorpusIndex
        ix = self.index
        if getattr(self, "_chunk_engine", None) is None:
            C = int(ix.n_chunks)
            # the engine's own device tensor when it holds the rows as they are (row-major layout): shared, not copied
            emb = self.engine._t["emb"] if getattr(self.engine, "scan_layout", 0) == 0 else ix.emb
            view = CorpusIndex(doc_ids=np.arange(C, dtype=np.int64), doc_off=np.arange(C + 1, dtype=np.int32),
                               chunk_ids=ix.chunk_ids, emb=emb, total_docs=C)
            self._chunk_engine = DeviceEngine(view, device=self.engine.device, max_queries=32,
                                              max_k=self.engine.max_k, rerank_max_docs=0)
            off = ix.doc_off.cpu().numpy() if hasattr(ix.doc_off, "cpu") else np.asarray(ix.doc_off)
            self._chunk_doc_off = off.astype(np.int64)
        qv = np.stack([self._embed(q, None if query_embeddings is None else query_embeddings[i])
                       for i, q in enumerate(queries if queries is not None else [None] * len(query_embeddings))])
        row, score, _, n = self._chunk_engine.dense_topk(qv, k=top_k, want_chunk=False)
        row, score, n = row.cpu().numpy(), score.cpu().numpy(), n.cpu().numpy()
        cid = ix.chunk_ids.cpu().numpy() if hasattr(ix.chunk_ids, "cpu") else np.asarray(ix.chunk_ids)
        out = []
        for r in range(len(qv)):
            rows = []
            for j in range(int(n[r])):
                c = int(row[r, j])
                i = int(np.searchsorted(self._chunk_doc_off, c, side="right") - 1)       # the chunk's document
                rows.append({"rank": j + 1, "doc_id": int(self._ids[i]), "chunk_id": int(cid[c]), "score": float(score[r, j]),
                             "url": ix.urls[i] if ix.urls is not None else None,
                             "title": ix.titles[i] if ix.titles is not None else None})
            out.append(rows)
        return out

    def quick_search(self, query=None, top_k=10, return_unique_docs=True, query_embedding=None, max_chunks_per_doc=0):
        return self.quick_search_batch([query], top_k, return_unique_docs,
                                       None if query_embedding is None else [query_embedding], max_chunks_per_doc)[0]

    # ------------------------------------------------------------------ live two-stage path
    def search_batch(self, queries, top_k=TOP_K_RETRIEVAL, query_embeddings=None, term_lists=None, query_ids=None):
        """-> per query the list of UI documents (search_api.py:110-130); [] when stage 1 finds nothing."""
        processed = [preprocess_query(q) for q in queries]
        if term_lists is None:
            term_lists = [self.bm25._tokenize(q) for q in processed]
        stage1 = [self.bm25._finish(r) for r in self.bm25.search_terms_batch(term_lists, top_k)]
        reqs, slot = [], []
        for i, res in enumerate(stage1):
            if res:
                reqs.append(dict(doc_ids=[str(r["doc_id"]) for r in res], similarities=[r["score"] for r in res],
                                 query=processed[i],
                                 query_embedding=None if query_embeddings is None else query_embeddings[i]))
                slot.append(i)
        out = [[] for _ in queries]
        if reqs:
            M = self.engine.rerank_max_docs
            for a in range(0, len(reqs), 32):
                chunk = reqs[a:a + 32]
                try:
                    resp = self.reranker.rerank_batch(chunk)
                except RerankNotFound:
                    resp = []
                    for rq in chunk:                       # isolate the query that has no chunk rows
                        try:
                            resp.append(self.reranker.rerank_batch([rq])[0])
                        except RerankNotFound:
                            resp.append(None)
                for i, rp in zip(slot[a:a + 32], resp):
                    if rp is None:
                        continue
                    docs = []
                    for rank, (d, w) in enumerate(zip(rp["document_scores"], rp["top_windows"]), start=1):
                        text = w.get("text", "")
                        docs.append({"query_id": None if query_ids is None else query_ids[i], "rank": rank,
                                     "url": d["url"], "score": d["similarity_score"],
                                     "title": d["title"] or "No Title",
                                     "snippet": (text[:200] + "..." if len(text) > 200 else text) or "No content available",
                                     "domain": extract_domain_topic(d["url"]), "doc_id": d["doc_id"]})
                    out[i] = docs
        return out

    def search(self, query, top_k=TOP_K_RETRIEVAL, query_embedding=None, terms=None, query_id=None):
        return self.search_batch([query], top_k, None if query_embedding is None else [query_embedding],
                                 None if terms is None else [terms], None if query_id is None else [query_id])[0]

    def batch_search(self, numbered_queries, query_embeddings=None, term_lists=None):
        """numbered_queries: [(query_num, text)] -> result entries with 'formatted_line' (search_api.py:276-292)."""
        docs = self.search_batch([q for _, q in numbered_queries], TOP_K_RETRIEVAL, query_embeddings, term_lists)
        out = []
        for (qn, _), ds in zip(numbered_queries, docs):
            for d in ds:
                out.append({"query_num": qn, "rank": d["rank"], "url": d["url"], "score": f"{d['score']:.3f}",
                            "formatted_line": format_result_line(qn, d["rank"], d["url"], d["score"])})
        return out

    def batch_search_to_file(self, queries_path, out_path, query_embeddings=None, term_lists=None):
        res = self.batch_search(read_queries_file(queries_path), query_embeddings, term_lists)
        with open(out_path, "w", encoding="utf-8") as f:
            for r in res:
                f.write(r["formatted_line"] + "\n")
        return len(res)
