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
from .reranker import Reranker
from .text import (LineFormatter, extract_domain, extract_domain_topic, format_result_line, preprocess_query,
                   read_queries_file)

TOP_K_RETRIEVAL = 1000     # config.py:13
TOP_K_RERANKING = 100      # config.py:14
RERANK_MAX_CHUNKS = 10     # reranker_api.py:58


class Retriever:
    def __init__(self, embedder=None, indexer=None, db_path=None, tokenizer=None, device=0, freeze_gc=False, **engine_kw):
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
        self._domains_bound = False
        self._formatter = None
        self._pinned = {}
        if freeze_gc:
            # The corpus side of a retriever is millions of long-lived Python objects (URL / title / text strings, the id maps):
            # every full garbage collection walks them -- tens of milliseconds, in the middle of a 10 ms batch.  They never
            # die, so they are moved to the permanent generation (gc.freeze), as a long-running server would do after start-up.
            import gc
            gc.collect()
            gc.freeze()

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
    # search_api.py:88-130 (single query) and :243-304 (batch): preprocess_query -> bm_25.search(top 1000) -> POST /rerank ->
    # formatted rows.  In the reference every arrow is a Python list of dicts (1000 per query, each with a 200-character
    # snippet or the full document text) and an HTTP/JSON hop.  Here stage 1 -> stage 2 -> diversification stay on the
    # device: msr_bm25_topk -> msr_rerank_gather -> msr_rerank_fuse -> msr_diversify, and only the FINAL <= ~100 rows per
    # query come back to the host, as arrays; URLs / titles / snippets are looked up for those rows only.
    def _doc_domains(self):
        """int32 [N]: domain id (extract_domain(url), reranker_api.py:170-176) of every document, -1 for documents that never
        appear in a response: no urlsDB row, or a NULL title / url / text (the reference's pydantic models reject them,
        :376-397).  Without URL metadata every url is "" -- ONE domain, as the facade's Reranker has it."""
        ix = self.index
        N = ix.n_docs
        if ix.urls is None:
            return np.zeros(N, np.int32)
        ids, out = {}, np.empty(N, np.int32)
        titles, texts = ix.titles, ix.texts
        for i, u in enumerate(ix.urls):
            if u is None or (titles is not None and titles[i] is None) or (texts is not None and texts[i] is None):
                out[i] = -1
            else:
                out[i] = ids.setdefault(extract_domain(u), len(ids))
        return out

    def _ensure_response_tables(self):
        if not self._domains_bound:
            self.engine.bind_doc_domains(self._doc_domains())
            self._domains_bound = True

    FINAL_COLS = 128           # columns of the final lists copied back per query (top_k = 100 + slack; a longer list -- more
    #                            than top_k "high" domains -- makes that chunk come back in full)

    def _enqueue_chunk(self, term_ids, qv, top_k, slot):
        """Device work of one chunk + the asynchronous copy of its final rows into pinned host buffers.  Only enqueues."""
        import torch
        eng, cfg = self.engine, self.reranker.cfg
        b = eng.bm25_topk(term_ids, k=top_k)
        cos, meta = eng.rerank_gather(qv, b[0], b[2], max_chunks=RERANK_MAX_CHUNKS)
        fused = eng.rerank_fuse(b[0], b[1], b[2], cos, meta, smoothing=cfg["smoothing"], max_chunks=RERANK_MAX_CHUNKS)
        fin = eng.diversify(fused, top_k=int(cfg["top_k"]), diversification=bool(cfg.get("diversification", False)))
        Qc, W = len(term_ids), min(self.FINAL_COLS, int(fin[0].shape[1]))
        pin = self._pinned.get(slot)
        if pin is None or pin[0].shape[0] < Qc or pin[0].shape[1] != W:
            rows = max(Qc, 256)
            pin = self._pinned[slot] = (torch.empty((rows, W), dtype=torch.int32, pin_memory=True),
                                        torch.empty((rows, W), dtype=torch.float64, pin_memory=True),
                                        torch.empty((rows, W), dtype=torch.int32, pin_memory=True),
                                        torch.empty((rows,), dtype=torch.int32, pin_memory=True))
        pin[0][:Qc].copy_(fin[0][:, :W], non_blocking=True)
        pin[1][:Qc].copy_(fin[1][:, :W], non_blocking=True)
        pin[2][:Qc].copy_(fin[3][:, :W], non_blocking=True)
        pin[3][:Qc].copy_(fin[4], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(eng.device))
        return fin, pin, ev, Qc, W

    @staticmethod
    def _collect_chunk(job):
        """Wait for a chunk's copies -> (doc, score, chunk row, n) numpy arrays of that chunk (copies: the pinned buffers are
        reused two chunks later)."""
        fin, pin, ev, Qc, W = job
        ev.synchronize()
        n = pin[3][:Qc].numpy().copy()
        S = int(n.max()) if Qc else 0
        if S > W:                                            # (rare) a list longer than the copied columns: this chunk in full
            return fin[0][:, :S].cpu().numpy(), fin[1][:, :S].cpu().numpy(), fin[3][:, :S].cpu().numpy(), n
        return pin[0][:Qc, :S].numpy().copy(), pin[1][:Qc, :S].numpy().copy(), pin[2][:Qc, :S].numpy().copy(), n

    def final_list_chunks(self, term_id_lists=None, query_vectors=None, top_k=TOP_K_RETRIEVAL, chunk=None, prepare=None,
                          n_queries=None):
        """The whole live path, chunk by chunk, on the device; yields (first query, doc index int32 [Qc, S], new_similarity
        float64 [Qc, S], winning chunk row int32 [Qc, S], n int32 [Qc]) per chunk of queries, rows in final rank order.
        Software-pipelined: while the GPU works on chunk i the host packs chunk i + 1 and the caller consumes chunk i - 1.
        prepare(a, b) (optional, with n_queries) -> (term id lists, vectors) of queries a .. b, evaluated just before the chunk
        is enqueued (text preprocessing inside the pipeline); otherwise term_id_lists / query_vectors hold all queries."""
        import torch
        eng = self.engine
        if top_k > eng.rerank_max_docs or top_k > eng.max_k:
            raise ValueError(f"top_k {top_k} exceeds the engine's max_k / rerank_max_docs ({eng.max_k} / {eng.rerank_max_docs})")
        self._ensure_response_tables()
        Q = len(term_id_lists) if prepare is None else int(n_queries)
        step = int(chunk or max(256, eng.max_queries))
        pending = None
        for i, a in enumerate(range(0, Q, step)):
            b = min(Q, a + step)
            ids, qv = prepare(a, b) if prepare is not None else (term_id_lists[a:b], query_vectors[a:b])
            qv = eng._dev(np.asarray(qv, np.float32) if not torch.is_tensor(qv) else qv, torch.float32).reshape(-1, 768)
            job = self._enqueue_chunk(ids, qv, top_k, i & 1)
            if pending is not None:
                yield (pending[0],) + self._collect_chunk(pending[1])
            pending = (a, job)
        if pending is not None:
            yield (pending[0],) + self._collect_chunk(pending[1])

    def final_lists(self, term_id_lists, query_vectors, top_k=TOP_K_RETRIEVAL, chunk=None):
        """-> host arrays (doc index int32 [Q, S], new_similarity float64 [Q, S], winning chunk row int32 [Q, S], n int32 [Q]);
        row q holds n[q] entries in final rank order (S = max n, normally the reranker's top_k = 100).  term_id_lists: per
        query its term ids (repeats allowed, unknown < 0); query_vectors [Q, 768]."""
        parts = list(self.final_list_chunks(term_id_lists, query_vectors, top_k, chunk))
        if not parts:
            z = np.zeros((0, 0), np.int32)
            return z, np.zeros((0, 0), np.float64), z, np.zeros(0, np.int32)
        S = max(p[1].shape[1] for p in parts)
        pad = lambda x, fill: x if x.shape[1] == S else np.concatenate(
            [x, np.full((x.shape[0], S - x.shape[1]), fill, x.dtype)], axis=1)
        return (np.concatenate([pad(p[1], -1) for p in parts]), np.concatenate([pad(p[2], -np.inf) for p in parts]),
                np.concatenate([pad(p[3], -1) for p in parts]), np.concatenate([p[4] for p in parts]))

    def _prepare(self, queries, query_embeddings, term_lists):
        processed = [preprocess_query(q) for q in queries]
        if term_lists is None:
            term_lists = [self.bm25._tokenize(q) for q in processed]
        ids = [self.index.term_ids(t) for t in term_lists]
        if isinstance(query_embeddings, np.ndarray) and query_embeddings.ndim == 2:
            qv = np.ascontiguousarray(query_embeddings, np.float32)             # (a matrix of vectors: taken as it is)
        else:
            qv = np.stack([self._embed(processed[i], None if query_embeddings is None else query_embeddings[i])
                           for i in range(len(queries))]) if len(queries) else np.zeros((0, 768), np.float32)
        return ids, qv

    def search_batch(self, queries, top_k=TOP_K_RETRIEVAL, query_embeddings=None, term_lists=None, query_ids=None):
        """-> per query the list of UI documents (search_api.py:110-130); [] when stage 1 finds nothing."""
        ids, qv = self._prepare(queries, query_embeddings, term_lists)
        doc, score, _, n = self.final_lists(ids, qv, top_k)
        ix = self.index
        out = []
        for q in range(len(queries)):
            rows = []
            qid = None if query_ids is None else query_ids[q]
            for r in range(int(n[q])):
                i = int(doc[q, r])
                url = ix.urls[i] if ix.urls is not None else ""
                title = ix.titles[i] if ix.titles is not None else ""
                text = ix.texts[i] if ix.texts is not None else ""
                rows.append({"query_id": qid, "rank": r + 1, "url": url, "score": float(score[q, r]),
                             "title": title or "No Title",
                             "snippet": (text[:200] + "..." if len(text) > 200 else text) or "No content available",
                             "domain": extract_domain_topic(url), "doc_id": str(int(self._ids[i]))})
            out.append(rows)
        return out

    def search(self, query, top_k=TOP_K_RETRIEVAL, query_embedding=None, terms=None, query_id=None):
        return self.search_batch([query], top_k, None if query_embedding is None else [query_embedding],
                                 None if terms is None else [terms], None if query_id is None else [query_id])[0]

    def batch_search(self, numbered_queries, query_embeddings=None, term_lists=None):
        """numbered_queries: [(query_num, text)] -> the result entries of search_api.py:276-292 ({query_num, rank, url, score,
        formatted_line}) as a BatchLines sequence: len / indexing / iteration give the reference's dicts, built on access;
        .text() / .write() produce all formatted lines natively (msr_format_lines) without building any."""
        ids, qv = self._prepare([q for _, q in numbered_queries], query_embeddings, term_lists)
        doc, score, _, n = self.final_lists(ids, qv, TOP_K_RETRIEVAL)
        if self._formatter is None:
            self._formatter = LineFormatter(self.index.urls, self.index.n_docs)
        return BatchLines([qn for qn, _ in numbered_queries], doc, score, n, self.index.urls, self._formatter)

    def batch_search_to_file(self, queries_path, out_path, query_embeddings=None, term_lists=None, chunk=None):
        """search_api.py:331-367: queries.txt -> one formatted line per result in out_path; -> number of lines.  Three things
        run side by side, chunk by chunk: this thread preprocesses / tokenises chunk i + 1 and enqueues it, the GPU ranks
        chunk i, a second host thread waits for chunk i - 1's final rows, formats them (native code, outside the interpreter
        lock) and writes them."""
        import sys
        nq = read_queries_file(queries_path)
        if self._formatter is None:
            self._formatter = LineFormatter(self.index.urls, self.index.n_docs)
        eng = self.engine
        if TOP_K_RETRIEVAL > eng.rerank_max_docs or TOP_K_RETRIEVAL > eng.max_k:
            raise ValueError(f"the batch path needs max_k / rerank_max_docs >= {TOP_K_RETRIEVAL}")
        self._ensure_response_tables()
        texts, nums = [q for _, q in nq], [n for n, _ in nq]
        sub = lambda x, a, b: None if x is None else x[a:b]
        step = int(chunk or max(256, eng.max_queries))
        # two threads hand the interpreter lock back and forth every fraction of a millisecond here; the default switch interval
        # (5 ms) is longer than a whole chunk takes
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(1e-4)
        try:
            return self._batch_to_file(nq, texts, nums, sub, step, out_path, query_embeddings, term_lists)
        finally:
            sys.setswitchinterval(old_switch)

    def _batch_to_file(self, nq, texts, nums, sub, step, out_path, query_embeddings, term_lists):
        from concurrent.futures import ThreadPoolExecutor
        import torch
        eng = self.engine
        with open(out_path, "wb") as f, ThreadPoolExecutor(max_workers=1) as pool:
            def consume(a, job):
                doc, score, _, n = self._collect_chunk(job)
                f.write(self._formatter.format(nums[a:a + len(n)], doc, score, n))
                return int(n.sum())
            futs = []
            for i, a in enumerate(range(0, len(nq), step)):
                b = min(len(nq), a + step)
                if i >= 2:
                    futs[i - 2].result()                      # its pinned buffers (slot i & 1) are free again
                ids, qv = self._prepare(texts[a:b], sub(query_embeddings, a, b), sub(term_lists, a, b))
                qv = eng._dev(qv, torch.float32).reshape(-1, 768)
                futs.append(pool.submit(consume, a, self._enqueue_chunk(ids, qv, TOP_K_RETRIEVAL, i & 1)))
            return sum(ft.result() for ft in futs)


class BatchLines:
    """The `results` list of /api/batch_search (search_api.py:276-292, 312-320) over the arrays the device returned: behaves
    like the reference's list of dicts (len, indexing, iteration, equality with a list), but an entry is built when it is
    asked for; the text of all lines comes from the native formatter in one call."""

    def __init__(self, query_nums, doc, score, n, urls, formatter):
        self.query_nums, self.doc, self.score, self.n, self.urls, self._fmt = query_nums, doc, score, n, urls, formatter
        self._start = np.zeros(len(n) + 1, np.int64)
        np.cumsum(n, out=self._start[1:])

    def __len__(self):
        return int(self._start[-1])

    def _entry(self, q, r):
        i = int(self.doc[q, r])
        url = (self.urls[i] if self.urls is not None else "") or ""
        sc = float(self.score[q, r])
        qn = self.query_nums[q]
        return {"query_num": qn, "rank": r + 1, "url": url, "score": f"{sc:.3f}",
                "formatted_line": format_result_line(qn, r + 1, url, sc)}

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[j] for j in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        q = int(np.searchsorted(self._start, k, side="right") - 1)
        return self._entry(q, int(k - self._start[q]))

    def __iter__(self):
        for q in range(len(self.n)):
            for r in range(int(self.n[q])):
                yield self._entry(q, r)

    def __eq__(self, other):
        return list(self) == list(other)

    def text(self) -> bytes:
        """All formatted lines, each ending in a newline (UTF-8)."""
        return self._fmt.format(self.query_nums, self.doc, self.score, self.n)

    def write(self, path):
        with open(path, "wb") as f:
            f.write(self.text())
