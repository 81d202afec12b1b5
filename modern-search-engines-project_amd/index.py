"""Host-side index container: the DuckDB tables of the reference, read ONCE into contiguous arrays.

Reference schema this mirrors (all read per query through SQL in the reference):
  urlsDB(id, url, title, text)                        crawler/databaseManagement.py:18-51
  bm25_doc_stats / bm25_term_freq / bm25_term_stats / bm25_corpus_stats    indexer/bm25_indexer.py:86-126
  chunks_optimized(chunk_id, doc_id, chunk_text), embeddings(chunk_id, FLOAT[768])   indexer/embedder.py:31-52

Layout: documents are numbered by the rank of their doc_id in ascending order ("dense index"); postings
are CSR by term with documents ascending inside a term; chunk rows are sorted by (doc, chunk_id) so a
document's chunks are one contiguous row range [doc_off[d], doc_off[d+1]).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

DIM = 768


def _np(x):
    if x is None:
        return None
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


@dataclass
class CorpusIndex:
    doc_ids: object                     # int64 [N] ascending
    doc_len: object = None              # int32 [N]
    term_off: object = None             # int64 [V+1]
    post_doc: object = None             # int32 [P]
    post_tf: object = None              # int32 [P]
    idf: object = None                  # float32 [V]  (as stored: REAL, may be negative)
    avgdl: float = 1.0                  # float32-rounded (REAL)
    total_docs: int = 0                 # N of the WHOLE corpus (replicated on shards)
    k1: float = 1.2                     # bm25_indexer.py:57
    b: float = 0.75
    vocab: Optional[Dict[str, int]] = None
    doc_off: object = None              # int32 [N+1]
    chunk_ids: object = None            # int64 [C]
    emb: object = None                  # float32 [C, 768] row-major (numpy array or torch tensor)
    urls: Optional[List[Optional[str]]] = None     # None entry => document not in urlsDB
    titles: Optional[List[Optional[str]]] = None
    texts: Optional[List[Optional[str]]] = None
    doc_base: int = 0                   # first global dense index of this shard
    row_base: int = 0                   # first global chunk row of this shard
    n_docs_global: int = 0
    _url_group: object = field(default=None, repr=False)

    # ------------------------------------------------------------------ basic properties
    @property
    def n_docs(self):
        return int(len(self.doc_ids))

    @property
    def n_terms(self):
        return 0 if self.term_off is None else int(len(self.term_off) - 1)

    @property
    def n_chunks(self):
        return 0 if self.doc_off is None else int(_np(self.doc_off[-1:])[0])

    def term_ids(self, terms):
        """Map term strings (or ints) to ids; unknown -> -1."""
        out = []
        for t in terms:
            if isinstance(t, (int, np.integer)):
                out.append(int(t))
            else:
                out.append(self.vocab.get(t, -1) if self.vocab else -1)
        return out

    def url_group(self):
        """int32[N]: id of the URL with its query string removed (reranker_api.py:43-46); -1 if the
        document has no urlsDB row.  Without URL metadata every document is its own group."""
        if self._url_group is None:
            if self.urls is None:
                # group ids are GLOBAL: a shard reloaded from a snapshot must not collide with its neighbours
                self._url_group = np.arange(self.doc_base, self.doc_base + self.n_docs, dtype=np.int32)
            else:
                ids, out = {}, np.empty(self.n_docs, np.int32)
                for i, u in enumerate(self.urls):
                    if u is None:
                        out[i] = -1
                    else:
                        key = u[: u.index("?")] if "?" in u else u
                        out[i] = ids.setdefault(key, len(ids))
                self._url_group = out
        return self._url_group

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def from_tables(postings, doc_len, idf, avgdl, total_docs=None, chunks=None, emb=None, urls_db=None,
                    k1=1.2, b=0.75):
        """Build from reference-style tables: postings {term: [(doc_id, tf)]}, doc_len {doc_id: len},
        idf {term: float|None}, chunks [(chunk_id, doc_id)], emb {chunk_id: vec} or array aligned with
        sorted chunks, urls_db {doc_id: (url, title, text)}.  Documents = keys of doc_len (the inner JOIN
        with bm25_doc_stats, bm25_indexer.py:443) united with chunk owners and urlsDB ids."""
        ids = set(int(d) for d in doc_len)
        if chunks:
            ids |= {int(d) for _, d in chunks}
        if urls_db:
            ids |= {int(d) for d in urls_db}
        doc_ids = np.array(sorted(ids), np.int64)
        rank = {int(d): i for i, d in enumerate(doc_ids)}
        vocab = {t: i for i, t in enumerate(postings)}
        term_off = np.zeros(len(vocab) + 1, np.int64)
        pd_, ptf = [], []
        for t, i in vocab.items():
            pl = sorted((rank[int(d)], int(tf)) for d, tf in postings[t] if int(d) in doc_len)
            pd_ += [p[0] for p in pl]
            ptf += [p[1] for p in pl]
            term_off[i + 1] = len(pd_)
        ix = CorpusIndex(doc_ids=doc_ids,
                         doc_len=np.array([doc_len.get(int(d), 0) for d in doc_ids], np.int32),
                         term_off=term_off, post_doc=np.array(pd_, np.int32), post_tf=np.array(ptf, np.int32),
                         idf=np.array([(idf[t] or 0.0) for t in vocab], np.float32),   # NULL -> 0.0 (:426)
                         avgdl=float(np.float32(avgdl)), total_docs=int(total_docs or len(doc_len)),
                         k1=k1, b=b, vocab=vocab)
        if chunks is not None:
            ch = sorted(((rank[int(d)], int(c)) for c, d in chunks))
            cnt = np.bincount(np.array([d for d, _ in ch], np.int64), minlength=len(doc_ids))
            ix.doc_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            ix.chunk_ids = np.array([c for _, c in ch], np.int64)
            if isinstance(emb, dict):
                ix.emb = np.stack([np.asarray(emb[c], np.float32) for c in ix.chunk_ids]) if len(ch) else np.zeros((0, DIM), np.float32)
            else:
                ix.emb = emb
        if urls_db is not None:
            ix.urls = [urls_db[int(d)][0] if int(d) in urls_db else None for d in doc_ids]
            ix.titles = [urls_db[int(d)][1] if int(d) in urls_db else None for d in doc_ids]
            ix.texts = [urls_db[int(d)][2] if int(d) in urls_db else None for d in doc_ids]
        ix.n_docs_global = ix.n_docs
        return ix

    # ------------------------------------------------------------------ snapshot (raw arrays)
    def save(self, path):
        arrs = dict(doc_ids=_np(self.doc_ids), doc_len=_np(self.doc_len), term_off=_np(self.term_off),
                    post_doc=_np(self.post_doc), post_tf=_np(self.post_tf), idf=_np(self.idf),
                    scalars=np.array([self.avgdl, self.total_docs, self.k1, self.b, self.doc_base,
                                      self.n_docs_global, self.row_base], np.float64),
                    url_group=np.asarray(self.url_group(), np.int32))
        if self.doc_off is not None:
            arrs.update(doc_off=_np(self.doc_off), chunk_ids=_np(self.chunk_ids), emb=_np(self.emb))
        if self.vocab is not None:
            arrs["vocab_terms"] = np.array(list(self.vocab.keys()), dtype=np.str_)
        np.savez(path, **{k: v for k, v in arrs.items() if v is not None})

    @staticmethod
    def load(path, mmap=False):
        z = np.load(path, mmap_mode="r" if mmap else None, allow_pickle=False)
        s = z["scalars"]
        ix = CorpusIndex(doc_ids=z["doc_ids"], doc_len=z["doc_len"], term_off=z["term_off"],
                         post_doc=z["post_doc"], post_tf=z["post_tf"], idf=z["idf"], avgdl=float(s[0]),
                         total_docs=int(s[1]), k1=float(s[2]), b=float(s[3]), doc_base=int(s[4]),
                         n_docs_global=int(s[5]), row_base=int(s[6]) if len(s) > 6 else 0)
        if "url_group" in z:
            ix._url_group = np.asarray(z["url_group"], np.int32)       # global group ids (a shard keeps the corpus-wide ones)
        if "doc_off" in z:
            ix.doc_off, ix.chunk_ids, ix.emb = z["doc_off"], z["chunk_ids"], z["emb"]
        if "vocab_terms" in z:
            ix.vocab = {str(t): i for i, t in enumerate(z["vocab_terms"])}
        return ix

    # ------------------------------------------------------------------ snapshot directory (memory-mappable)
    _ARRAYS = ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf", "doc_off", "chunk_ids", "emb")

    def save_dir(self, path, block_rows=1 << 18, skip=()):
        """Export format for the reference's tables: one little-endian `.npy` per array + `meta.json`
        (+ `docs.jsonl` with url / title / text per document when present).  Every array is written through a
        memory map in blocks, so a 15 GB embedding matrix that lives on the GPU is never held twice on the host.
        `load_dir(..., mmap=True)` maps the files back without reading them; DeviceEngine then streams them to HBM
        through pinned staging buffers (engine.stream_to_device)."""
        import json
        import os
        os.makedirs(path, exist_ok=True)
        present = []
        for name in self._ARRAYS:
            a = getattr(self, name)
            if a is None:
                continue
            present.append(name)
            if name in skip:                                   # already written in place (from_duckdb(snapshot_dir=...))
                continue
            shape = tuple(int(x) for x in a.shape)
            dt = np.dtype(str(a.dtype).replace("torch.", "")) if hasattr(a, "detach") else np.asarray(a[:0]).dtype
            out = np.lib.format.open_memmap(os.path.join(path, name + ".npy"), mode="w+", dtype=dt, shape=shape)
            step = max(1, block_rows if len(shape) > 1 else block_rows * DIM)
            for s0 in range(0, shape[0], step):
                out[s0:s0 + step] = _np(a[s0:s0 + step])
            out.flush()
            del out
        np.save(os.path.join(path, "url_group.npy"), np.asarray(self.url_group(), np.int32))   # GLOBAL group ids
        meta = dict(format="msretr-snapshot-1", arrays=present, avgdl=float(self.avgdl), total_docs=int(self.total_docs),
                    k1=float(self.k1), b=float(self.b), doc_base=int(self.doc_base), row_base=int(self.row_base),
                    n_docs_global=int(self.n_docs_global), vocab=list(self.vocab.keys()) if self.vocab is not None else None,
                    has_docs=self.urls is not None)
        with open(os.path.join(path, "meta.json"), "w", encoding="utf-8") as f:
            json.dump(meta, f, ensure_ascii=False)
        if self.urls is not None:
            with open(os.path.join(path, "docs.jsonl"), "w", encoding="utf-8") as f:
                for i in range(self.n_docs):
                    f.write(json.dumps([self.urls[i], self.titles[i] if self.titles else None,
                                        self.texts[i] if self.texts else None], ensure_ascii=False) + "\n")

    @staticmethod
    def load_dir(path, mmap=True):
        import json
        import os
        with open(os.path.join(path, "meta.json"), encoding="utf-8") as f:
            meta = json.load(f)
        if meta.get("format") != "msretr-snapshot-1":
            raise ValueError(f"{path}: not an msretr snapshot directory")
        arrs = {n: np.load(os.path.join(path, n + ".npy"), mmap_mode="r" if mmap else None, allow_pickle=False)
                for n in meta["arrays"]}
        ix = CorpusIndex(doc_ids=arrs["doc_ids"], avgdl=meta["avgdl"], total_docs=meta["total_docs"], k1=meta["k1"],
                         b=meta["b"], doc_base=meta["doc_base"], row_base=meta["row_base"],
                         n_docs_global=meta["n_docs_global"])
        for n in CorpusIndex._ARRAYS[1:]:
            if n in arrs:
                setattr(ix, n, arrs[n])
        if meta.get("vocab") is not None:
            ix.vocab = {t: i for i, t in enumerate(meta["vocab"])}
        if os.path.exists(os.path.join(path, "url_group.npy")):
            # the groups were numbered over the WHOLE corpus before sharding: recomputing them from a shard's own
            # URLs would make unrelated documents of different shards share an id (rerank dedup keeps MIN(doc))
            ix._url_group = np.load(os.path.join(path, "url_group.npy"), allow_pickle=False)
        if meta.get("has_docs"):
            ix.urls, ix.titles, ix.texts = [], [], []
            with open(os.path.join(path, "docs.jsonl"), encoding="utf-8") as f:
                for line in f:
                    u, t, x = json.loads(line)
                    ix.urls.append(u); ix.titles.append(t); ix.texts.append(x)
        return ix

    # ------------------------------------------------------------------ DuckDB (the reference's store)
    @staticmethod
    def from_duckdb(db_path, with_text=True, connect=None, snapshot_dir=None, block_docs=1 << 16):
        """Read the reference's tables ONCE into contiguous arrays, column-wise (no per-row Python work on the big tables):
          * postings: the term -> id join and both orderings happen in SQL, the three numeric columns come back through
            fetchnumpy(); document ids are mapped to dense indices with one np.searchsorted; CSR offsets with np.bincount;
          * embeddings: fetched block-wise by document-id range (the reference's idx on chunks_optimized.doc_id serves it)
            and written straight into the destination matrix -- a `.npy` memory map under `snapshot_dir` when given, so a
            15 GB matrix is never held as Python lists and the snapshot (save_dir format) is complete when this returns.
        `duckdb` is not installable in the build container: the loader is written against the DDL cited in the module
        docstring and is exercised there through `connect`, a callable that returns a DuckDB-style connection
        (`.execute(sql, params).fetchall()` / `.fetchnumpy()`); tests/test_host_logic.py passes a sqlite3 adapter over
        tables created with the reference's own column names.  Default: duckdb, read-only (search_api.py:32,48)."""
        import os
        if connect is None:
            import duckdb  # noqa: deliberately unguarded: fails loudly where duckdb is absent
            con = duckdb.connect(db_path, read_only=True)
        else:
            con = connect(db_path)
        docs = con.execute("SELECT doc_id, doc_length FROM bm25_doc_stats ORDER BY doc_id").fetchnumpy()
        stats = dict(con.execute("SELECT stat_name, stat_value FROM bm25_corpus_stats").fetchall())
        tcols = con.execute("SELECT term, idf_score FROM bm25_term_stats ORDER BY term").fetchnumpy()
        term_names = [str(t) for t in np.asarray(tcols["term"]).tolist()]
        vocab = {t: i for i, t in enumerate(term_names)}
        idf_raw = np.ma.filled(np.ma.masked_invalid(np.ma.array(
            [np.nan if v is None else v for v in np.asarray(tcols["idf_score"], dtype=object).tolist()], dtype=np.float64)), 0.0)
        ch = con.execute("SELECT c.chunk_id, c.doc_id FROM chunks_optimized c ORDER BY c.doc_id, c.chunk_id").fetchnumpy()
        ch_doc = np.asarray(ch["doc_id"], np.int64)
        url_ids = np.asarray(con.execute("SELECT id FROM urlsDB ORDER BY id").fetchnumpy()["id"], np.int64)
        doc_stat_ids = np.asarray(docs["doc_id"], np.int64)
        all_ids = np.union1d(np.union1d(doc_stat_ids, ch_doc), url_ids)
        n_docs = len(all_ids)
        doc_len = np.zeros(n_docs, np.int32)
        doc_len[np.searchsorted(all_ids, doc_stat_ids)] = np.asarray(docs["doc_length"]).astype(np.int32)
        # postings: (term rank, doc_id, freq) already in (term, doc) order; only documents with a doc_stats row (:443)
        tf = con.execute(
            "SELECT ts.rn AS term_id, tf.doc_id AS doc_id, tf.freq AS freq FROM bm25_term_freq tf "
            "JOIN (SELECT term, ROW_NUMBER() OVER (ORDER BY term) - 1 AS rn FROM bm25_term_stats) ts ON tf.term = ts.term "
            "JOIN bm25_doc_stats ds ON tf.doc_id = ds.doc_id ORDER BY ts.rn, tf.doc_id").fetchnumpy()
        t_id = np.asarray(tf["term_id"], np.int64)
        term_off = np.zeros(len(vocab) + 1, np.int64)
        term_off[1:] = np.cumsum(np.bincount(t_id, minlength=len(vocab)))
        post_doc = np.searchsorted(all_ids, np.asarray(tf["doc_id"], np.int64)).astype(np.int32)
        post_tf = np.asarray(tf["freq"]).astype(np.int32)
        del tf, t_id
        # chunk layout
        ch_rank = np.searchsorted(all_ids, ch_doc)
        cnt = np.bincount(ch_rank, minlength=n_docs)
        doc_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        n_chunks = int(doc_off[-1])
        # embeddings, block by block of documents, straight into the destination
        if snapshot_dir is not None:
            os.makedirs(snapshot_dir, exist_ok=True)
            emb = np.lib.format.open_memmap(os.path.join(snapshot_dir, "emb.npy"), mode="w+", dtype=np.float32, shape=(n_chunks, DIM))
        else:
            emb = np.empty((n_chunks, DIM), np.float32)
        for b0 in range(0, n_docs, block_docs):
            b1 = min(n_docs, b0 + block_docs)
            r0, r1 = int(doc_off[b0]), int(doc_off[b1])
            if r1 == r0:
                continue
            col = con.execute(
                "SELECT e.embedding AS embedding FROM embeddings e JOIN chunks_optimized c ON e.chunk_id = c.chunk_id "
                "WHERE c.doc_id >= ? AND c.doc_id <= ? ORDER BY c.doc_id, c.chunk_id",
                (int(all_ids[b0]), int(all_ids[b1 - 1]))).fetchnumpy()["embedding"]
            block = np.asarray(col)
            if block.dtype == object:                             # one array per row (LIST / ARRAY columns)
                block = np.stack(block.tolist()) if len(block) else np.zeros((0, DIM), np.float32)
            if block.shape != (r1 - r0, DIM):
                raise ValueError(f"embeddings of documents {all_ids[b0]}..{all_ids[b1 - 1]}: got {block.shape}, chunk table says {(r1 - r0, DIM)}")
            emb[r0:r1] = block.astype(np.float32, copy=False)
        # urlsDB text columns: python strings by nature; one pass, positions by searchsorted
        url_rows = con.execute("SELECT id, url, title, text FROM urlsDB ORDER BY id").fetchall() if with_text else \
            con.execute("SELECT id, url, title, NULL FROM urlsDB ORDER BY id").fetchall()
        urls, titles, texts = [None] * n_docs, [None] * n_docs, [None] * n_docs
        pos = np.searchsorted(all_ids, np.fromiter((r[0] for r in url_rows), np.int64, len(url_rows)))
        for p, r in zip(pos.tolist(), url_rows):
            urls[p], titles[p], texts[p] = r[1], r[2], r[3]
        ix = CorpusIndex(doc_ids=all_ids, doc_len=doc_len, term_off=term_off, post_doc=post_doc, post_tf=post_tf,
                         idf=idf_raw.astype(np.float32),                       # NULL -> 0.0 (:426)
                         avgdl=float(np.float32(stats.get("avg_doc_length", 1.0))),
                         total_docs=int(stats.get("total_docs", len(doc_stat_ids))), vocab=vocab,
                         doc_off=doc_off, chunk_ids=np.asarray(ch["chunk_id"], np.int64), emb=emb,
                         urls=urls, titles=titles, texts=texts)
        ix.n_docs_global = ix.n_docs
        if snapshot_dir is not None:
            emb.flush()
            ix.save_dir(snapshot_dir, skip=("emb",))              # the matrix is already in place
        return ix

    # ------------------------------------------------------------------ doc-range sharding
    def shard_bounds(self, world):
        """Document boundaries of `world` shards balanced by chunk count (dense cost); falls back to
        document count when no chunks are present."""
        N = self.n_docs
        if self.doc_off is not None and self.n_chunks > 0:
            off = _np(self.doc_off).astype(np.int64)
            want = (off[-1] * np.arange(1, world)) // world
            cuts = np.searchsorted(off, want, side="left")
        else:
            cuts = (N * np.arange(1, world)) // world
        b = np.concatenate([[0], np.minimum(cuts, N), [N]]).astype(np.int64)
        return np.maximum.accumulate(b)

    def shard(self, rank, world):
        """Shard `rank` of `world`: its documents, its slice of every posting list, its chunk rows.
        idf / avgdl / total_docs stay GLOBAL so scores are bit-identical to the unsharded index.
        Works on numpy arrays and on torch tensors (the 1 M-document corpus is sharded on the GPU)."""
        b = self.shard_bounds(world)
        d0, d1 = int(b[rank]), int(b[rank + 1])
        sub = CorpusIndex(doc_ids=self.doc_ids[d0:d1], avgdl=self.avgdl, total_docs=self.total_docs,
                          k1=self.k1, b=self.b, vocab=self.vocab, doc_base=self.doc_base + d0,
                          n_docs_global=self.n_docs_global or self.n_docs)
        if self.term_off is not None:
            pdoc = self.post_doc
            keep = (pdoc >= d0) & (pdoc < d1)
            if hasattr(pdoc, "detach"):
                import torch
                csum = torch.zeros(pdoc.numel() + 1, dtype=torch.int64, device=pdoc.device)
                csum[1:] = torch.cumsum(keep, 0)
                sub.term_off = csum[self.term_off.to(torch.int64)]
                sub.post_doc = (pdoc[keep] - d0).to(torch.int32)
            else:
                csum = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
                sub.term_off = csum[np.asarray(self.term_off)]
                sub.post_doc = (pdoc[keep] - d0).astype(np.int32)
            sub.post_tf = self.post_tf[keep]
            sub.doc_len = self.doc_len[d0:d1]
            sub.idf = self.idf
        if self.doc_off is not None:
            c0, c1 = int(self.doc_off[d0]), int(self.doc_off[d1])
            sub.doc_off = self.doc_off[d0:d1 + 1] - c0
            sub.chunk_ids = self.chunk_ids[c0:c1] if self.chunk_ids is not None else None
            sub.emb = self.emb[c0:c1] if self.emb is not None else None
            sub.row_base = c0
        for name in ("urls", "titles", "texts"):
            v = getattr(self, name)
            if v is not None:
                setattr(sub, name, v[d0:d1])
        sub._url_group = self.url_group()[d0:d1]            # group ids stay GLOBAL (dedup works across shards)
        return sub
