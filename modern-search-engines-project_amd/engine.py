"""DeviceEngine: the index resident in HBM + the libmsretr handle.

PyTorch is plumbing here: it owns the device allocations and the stream; every computation on the path is
a HIP kernel in csrc/ reached through the C ABI (include/msretr.h).  There is no fallback: without a GPU
or without the built library, construction raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from .index import DIM, CorpusIndex

RERANK_DEFAULTS = dict(smoothing=0.15, max_boost=0.1, max_decay=0.05, max_chunks=10)   # reranker/config.yaml:28,
#                                                                                        reranker_api.py:58,317-318


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


STREAM_MIN_BYTES = 64 << 20
_NP_OF = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64}


def stream_to_device(arr, device, block_bytes=128 << 20):
    """numpy array (typically a read-only memory map of a snapshot file, index.CorpusIndex.load_dir) -> device tensor,
    block by block through two pinned staging buffers: while block i travels to HBM (async copy on the current
    stream) block i + 1 is read from the file into the other buffer.  The host never holds more than two blocks."""
    dev = torch.device(device)
    out = torch.empty(arr.shape, dtype=torch.from_numpy(np.empty(0, arr.dtype)).dtype, device=dev)
    if arr.size == 0:
        return out
    flat_out = out.view(-1)
    row = int(np.prod(arr.shape[1:], dtype=np.int64)) if arr.ndim > 1 else 1
    rows_per_block = max(1, block_bytes // max(1, row * arr.itemsize))
    pinned = dev.type == "cuda"
    stage = [torch.empty(rows_per_block * row, dtype=out.dtype, pin_memory=pinned) for _ in range(2)]
    done = [None, None]
    for b, r0 in enumerate(range(0, arr.shape[0], rows_per_block)):
        r1 = min(arr.shape[0], r0 + rows_per_block)
        buf = stage[b & 1]
        if done[b & 1] is not None:
            done[b & 1].synchronize()                    # the copy that last used this buffer has finished
        n = (r1 - r0) * row
        np.copyto(buf.numpy()[:n].reshape((r1 - r0,) + tuple(arr.shape[1:])), arr[r0:r1])
        flat_out[r0 * row:r0 * row + n].copy_(buf[:n], non_blocking=pinned)
        if pinned:
            done[b & 1] = torch.cuda.Event()
            done[b & 1].record(torch.cuda.current_stream(dev))
    if pinned:
        torch.cuda.current_stream(dev).synchronize()
    return out


class DeviceEngine:
    def __init__(self, index: CorpusIndex, device=0, max_queries=32, max_k=1000, rerank_max_docs=1000,
                 scan_layout=0, scan_variant=0, row_copy=True):
        """row_copy=False: MSR_CFG_NO_ROW_COPY -- the 256-query pass reads the row-major matrix instead of an engine-owned
        fragment-order copy of it (half the embedding footprint, a slower pass, the same results)."""
        if not torch.cuda.is_available():
            raise _abi.MsrError(-102, "no GPU visible: the retrieval path runs on MI355X only (no CPU fallback)")
        self.lib = _abi.load()
        self.index = index
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.max_k = int(max_k)
        self.max_queries = int(max_queries)
        self.scan_layout = int(scan_layout)
        self.rerank_max_docs = int(rerank_max_docs)
        cfg = _abi.MsrConfig(C.sizeof(_abi.MsrConfig), self.device.index or 0, DIM, int(max_queries), int(max_k),
                             int(rerank_max_docs), int(scan_layout), int(scan_variant),
                             0 if row_copy else _abi.MSR_CFG_NO_ROW_COPY)
        self.handle = C.c_void_p()
        rc = self.lib.msr_create(C.byref(cfg), C.byref(self.handle))
        if rc != 0:
            raise _abi.MsrError(rc, (self.lib.msr_last_error(None) or b"?").decode())
        self._t = {}          # device tensors that the engine borrows: keep them alive
        self._bind(index)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype):
        if x is None:
            return None
        if torch.is_tensor(x):
            return x.to(device=self.device, dtype=dtype).contiguous()
        if isinstance(x, np.ndarray) and x.nbytes >= STREAM_MIN_BYTES and x.dtype == _NP_OF[dtype]:
            return stream_to_device(x, self.device)      # large (memory-mapped) arrays: pinned double buffer
        x = np.ascontiguousarray(x)
        if not x.flags.writeable:                            # small read-only memory maps: torch wants a writable source
            x = x.copy()
        return torch.as_tensor(x).to(device=self.device, dtype=dtype)

    def _check(self, rc):
        _abi.check(self.handle, rc)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            torch.cuda.synchronize(self.device)
            self.lib.msr_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bind(self, ix):
        t = self._t
        with torch.cuda.device(self.device):
            if ix.term_off is not None and ix.n_terms > 0:
                t["term_off"] = self._dev(ix.term_off, torch.int64)
                t["post_doc"] = self._dev(ix.post_doc, torch.int32)
                t["post_tf"] = self._dev(ix.post_tf, torch.int32)
                t["doc_len"] = self._dev(ix.doc_len, torch.int32)
                t["idf"] = self._dev(ix.idf, torch.float32)
                self._check(self.lib.msr_bind_postings(
                    self.handle, _ptr(t["term_off"]), ix.n_terms, _ptr(t["post_doc"]), _ptr(t["post_tf"]),
                    int(t["post_doc"].numel()), _ptr(t["doc_len"]), ix.n_docs, _ptr(t["idf"]),
                    C.c_float(ix.avgdl), C.c_double(ix.k1), C.c_double(ix.b), self._stream()))
            if ix.doc_off is not None and ix.emb is not None:
                t["doc_off"] = self._dev(ix.doc_off, torch.int32)
                emb = self._dev(ix.emb, torch.float32)
                n_chunks = int(emb.shape[0])
                inv = None
                if self.scan_layout == 1:
                    nrm = torch.linalg.vector_norm(emb, dim=1)
                    inv = (1.0 / torch.where(nrm == 0, torch.ones_like(nrm), nrm)).contiguous()
                    pad = (n_chunks + 15) // 16 * 16
                    tiled = torch.empty((pad, DIM), dtype=torch.float32, device=self.device)
                    self._check(self.lib.msr_interleave_rows(self.handle, _ptr(emb), n_chunks, _ptr(tiled), self._stream()))
                    torch.cuda.synchronize(self.device)
                    del emb
                    emb = tiled
                t["emb"], t["inv_norm"] = emb, inv
                self._check(self.lib.msr_bind_chunks(self.handle, _ptr(emb), n_chunks, _ptr(t["doc_off"]),
                                                     ix.n_docs, _ptr(inv), self._stream()))
                t["url_group"] = self._dev(ix.url_group(), torch.int32)
                self._check(self.lib.msr_bind_doc_meta(self.handle, _ptr(t["url_group"]), ix.n_docs, self._stream()))
            torch.cuda.synchronize(self.device)

    def scan_arith(self):
        """'f32' (exact f32 MFMA) or 'f16x2' (f32 rows split into two f16 pieces, f32 accumulation)."""
        return {0: "f32", 1: "f16x2"}.get(self.lib.msr_scan_arith(self.handle), "none")

    # ------------------------------------------------------------------ stage 1
    def batch_width(self):
        """Queries served by one bf16 sweep in dense_topk_batched (128 or 64; -1 before enable_bf16)."""
        return int(self.lib.msr_batch_width(self.handle))

    def batch_gemm_ok(self):
        """True if dense_topk_batched runs batches of more than 128 queries as the tiled matrix-core GEMM."""
        return bool(self.lib.msr_batch_gemm_ok(self.handle))

    def scan_width(self):
        """Most queries one pass over the embedding matrix serves in dense_topk (256 / 128: streaming pass, 64: K-split sweep,
        else 32)."""
        return int(self.lib.msr_scan_width(self.handle))

    def row_copy_state(self):
        """'none' (not applicable), 'built', 'declined' (row_copy=False) or 'alloc_failed' (fell back to the row-major matrix)."""
        return {0: "none", 1: "built", 2: "declined", 3: "alloc_failed"}.get(self.lib.msr_row_copy_state(self.handle), "?")

    def row_image_state(self):
        """The f16 image of the rows that launches of several 256-query groups read (max_queries >= 512): 'none', 'built',
        'declined' or 'alloc_failed' (those launches then convert the f32 rows in registers, like a single-group launch)."""
        return {0: "none", 1: "built", 2: "declined", 3: "alloc_failed"}.get(self.lib.msr_row_image_state(self.handle), "?")

    def owned_bytes(self):
        """Device bytes the handle owns (scratch, tables and copies built at bind); the bound index tensors are not included."""
        return int(self.lib.msr_owned_bytes(self.handle))

    def dense_path(self):
        """Queries per pass of the kernel the most recent dense_topk call ran (256 / 128 / 64 / 32; 0 before the first)."""
        return int(self.lib.msr_dense_path(self.handle))

    def pack_queries(self, term_lists):
        """list of term-id lists (repeats allowed, any unknown id < 0) -> device CSR of UNIQUE terms in
        first-occurrence order with their query frequencies (bm25_indexer.py:405-409)."""
        off, terms, qtf = [0], [], []
        for tl in term_lists:
            cnt = {}
            for t in tl:
                cnt[int(t)] = cnt.get(int(t), 0) + 1
            if len(cnt) > 64:
                raise ValueError("a query may hold at most 64 unique terms (MSR_MAX_QUERY_TERMS)")
            for t, c in cnt.items():
                terms.append(t)
                qtf.append(c)
            off.append(len(terms))
        mk = lambda a: torch.tensor(a if a else [0], dtype=torch.int32, device=self.device)
        return mk(off), mk(terms), mk(qtf), len(term_lists)

    def bm25_topk(self, term_lists, k=1000, min_score=0.0, packed=None):
        """-> (doc index int32 [Q, k], score float64 [Q, k], n int32 [Q]) device tensors."""
        q_off, q_terms, q_qtf, Q = packed if packed is not None else self.pack_queries(term_lists)
        out_doc = torch.empty((Q, k), dtype=torch.int32, device=self.device)
        out_score = torch.empty((Q, k), dtype=torch.float64, device=self.device)
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_bm25_topk(self.handle, _ptr(q_off), _ptr(q_terms), _ptr(q_qtf), Q, k,
                                           C.c_double(min_score), _ptr(out_doc), _ptr(out_score), _ptr(out_n),
                                           self._stream()))
        return out_doc, out_score, out_n

    # ------------------------------------------------------------------ stage 2 (full scan)
    def dense_topk(self, qvec, k=100, max_chunks_per_doc=0, want_chunk=True):
        """qvec float32 [Q, 768] (not normalised) -> (doc [Q,k] i32, score [Q,k] f32, chunk row [Q,k] i32, n [Q])."""
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        Q = int(q.shape[0])
        out_doc = torch.empty((Q, k), dtype=torch.int32, device=self.device)
        out_score = torch.empty((Q, k), dtype=torch.float32, device=self.device)
        out_chunk = torch.empty((Q, k), dtype=torch.int32, device=self.device) if want_chunk else None
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_dense_topk(self.handle, _ptr(q), Q, k, int(max_chunks_per_doc), _ptr(out_doc),
                                            _ptr(out_score), _ptr(out_chunk), _ptr(out_n), self._stream()))
        return out_doc, out_score, out_chunk, out_n

    def dense_split_max(self, k=100):
        """Most queries one dense_begin / dense_end pair takes (0: this engine cannot split the dense call)."""
        return int(self.lib.msr_dense_split_max(self.handle, int(k)))

    def dense_begin(self, qvec, k=100, k_part=None):
        """First half of dense_topk for a doc-sharded index (msr_dense_topk_begin): -> part float32 [Q] (device): a cosine that
        k_part documents of this shard reach exactly.  The caller takes the minimum over the shards (k_part = ceil(k / shards))
        and hands it to dense_end."""
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        Q = int(q.shape[0])
        part = torch.empty((Q,), dtype=torch.float32, device=self.device)
        self._check(self.lib.msr_dense_topk_begin(self.handle, _ptr(q), Q, int(k), int(k_part or k), _ptr(part), self._stream()))
        return part

    def dense_end(self, Q, k=100, bound=None, want_chunk=True, out=None):
        """Second half (msr_dense_topk_end): bound float32 [Q] (device) or None -> (doc, score, chunk row, n) as dense_topk; with
        a bound n may be < k -- every document this shard can contribute to the global top-k.  out: optional (doc, score,
        chunk, n) contiguous tensors of those shapes to write into (row slices of a larger result)."""
        if out is not None:
            out_doc, out_score, out_chunk, out_n = out
            assert tuple(out_doc.shape) == (Q, k) and out_doc.is_contiguous() and out_score.is_contiguous() and out_n.is_contiguous()
        else:
            out_doc = torch.empty((Q, k), dtype=torch.int32, device=self.device)
            out_score = torch.empty((Q, k), dtype=torch.float32, device=self.device)
            out_chunk = torch.empty((Q, k), dtype=torch.int32, device=self.device) if want_chunk else None
            out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_dense_topk_end(self.handle, int(Q), int(k), _ptr(bound), _ptr(out_doc), _ptr(out_score),
                                                _ptr(out_chunk), _ptr(out_n), self._stream()))
        return out_doc, out_score, out_chunk, out_n

    def enable_bf16(self):
        """Build the bf16 copy of the embeddings used by dense_topk_batched (+7.7 GB at 5 M chunks)."""
        self._check(self.lib.msr_enable_bf16(self.handle, self._stream()))
        torch.cuda.synchronize(self.device)

    def dense_topk_batched(self, qvec, k=100, max_chunks_per_doc=0, want_chunk=True):
        """Throughput variant of dense_topk: bf16 candidate sweep (up to 128 queries per sweep) + exact f32 rescoring.
        Same outputs; queries whose candidate set overflowed are rerun on the exact f32 scan."""
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        Q = int(q.shape[0])
        out_doc = torch.empty((Q, k), dtype=torch.int32, device=self.device)
        out_score = torch.empty((Q, k), dtype=torch.float32, device=self.device)
        out_chunk = torch.empty((Q, k), dtype=torch.int32, device=self.device) if want_chunk else None
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_dense_topk_bf16(self.handle, _ptr(q), Q, k, int(max_chunks_per_doc), _ptr(out_doc),
                                                 _ptr(out_score), _ptr(out_chunk), _ptr(out_n), self._stream()))
        bad = torch.nonzero(out_n < 0).flatten()
        if bad.numel():                                   # (syncs; rare) exact rerun of the overflowed queries
            d, s, c, n = self.dense_topk(q[bad], k=k, max_chunks_per_doc=max_chunks_per_doc, want_chunk=want_chunk)
            out_doc[bad], out_score[bad], out_n[bad] = d, s, n
            if want_chunk:
                out_chunk[bad] = c
        return out_doc, out_score, out_chunk, out_n

    # ------------------------------------------------------------------ rerank / fuse
    def rerank(self, qvec, cand_doc, cand_bm25, cand_n, **params):
        """cand_doc int32 [Q, M] dense indices, cand_bm25 float64 [Q, M], cand_n int32 [Q]
        -> (doc, new_similarity f64, normalised bm25 f64, chunk row, n, rows) device tensors [Q, M] / [Q]."""
        p = dict(RERANK_DEFAULTS)
        p.update(params)
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        cand_doc = self._dev(cand_doc, torch.int32)
        cand_bm25 = self._dev(cand_bm25, torch.float64)
        cand_n = self._dev(cand_n, torch.int32)
        Q, M = int(cand_doc.shape[0]), int(cand_doc.shape[1])
        prm = _abi.MsrRerankParams(p["smoothing"], p["max_boost"], p["max_decay"], int(p["max_chunks"]), 0)
        mk = lambda dt: torch.empty((Q, M), dtype=dt, device=self.device)
        out_doc, out_score, out_orig, out_chunk = mk(torch.int32), mk(torch.float64), mk(torch.float64), mk(torch.int32)
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        out_rows = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_rerank(self.handle, _ptr(q), Q, _ptr(cand_doc), _ptr(cand_bm25), _ptr(cand_n), M,
                                        C.byref(prm), _ptr(out_doc), _ptr(out_score), _ptr(out_orig),
                                        _ptr(out_chunk), _ptr(out_n), _ptr(out_rows), self._stream()))
        return out_doc, out_score, out_orig, out_chunk, out_n, out_rows

    def rerank_gather(self, qvec, cand_doc_global, cand_n, doc_base=0, row_base=0, max_chunks=10, out=None):
        """Shard-local half of rerank: (cos float32 [Q, M, 10], meta int32 [Q, M, 3]); zeros for foreign docs.
        out: optional (cos, meta) contiguous tensors of those shapes to write into (views of an exchange buffer)."""
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        cand = self._dev(cand_doc_global, torch.int32)
        cn = self._dev(cand_n, torch.int32)
        Q, M = int(cand.shape[0]), int(cand.shape[1])
        if out is not None:
            cos, meta = out
            assert cos.is_contiguous() and meta.is_contiguous() and cos.dtype == torch.float32 and meta.dtype == torch.int32
            assert tuple(cos.shape) == (Q, M, _abi.MSR_RERANK_MAX_CHUNKS) and tuple(meta.shape) == (Q, M, 3)
        else:
            cos = torch.empty((Q, M, _abi.MSR_RERANK_MAX_CHUNKS), dtype=torch.float32, device=self.device)
            meta = torch.empty((Q, M, 3), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_rerank_gather(self.handle, _ptr(q), Q, _ptr(cand), _ptr(cn), M, int(doc_base),
                                               int(row_base), int(max_chunks), _ptr(cos), _ptr(meta), self._stream()))
        return cos, meta

    def rerank_gather_blocks(self, qvec, cand_doc_global, cand_n, blocks, queries_per_block, doc_base=0, row_base=0, max_chunks=10):
        """rerank_gather for ALL queries in one launch, written into `blocks` (int32 [n_blocks, block_words], contiguous: the
        send buffer of the all-to-all): block b = [cos of queries b * qpb .. | their meta | padding] (msr_rerank_gather_blocks)."""
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        cand = self._dev(cand_doc_global, torch.int32)
        cn = self._dev(cand_n, torch.int32)
        Q, M = int(cand.shape[0]), int(cand.shape[1])
        assert blocks.is_contiguous() and blocks.dtype == torch.int32 and blocks.dim() == 2
        assert int(blocks.shape[0]) * int(queries_per_block) >= Q
        self._check(self.lib.msr_rerank_gather_blocks(self.handle, _ptr(q), Q, _ptr(cand), _ptr(cn), M, int(doc_base), int(row_base),
                                                      int(max_chunks), _ptr(blocks), int(queries_per_block), int(blocks.shape[1]),
                                                      self._stream()))

    def rerank_combine(self, cos_parts, meta_parts, nq):
        """Join of the gathered halves: cos_parts float32 [G, Qs, M, 10] and meta_parts int32 [G, Qs, M, 3] are views of ONE
        receive buffer (the same stride between parts, each part contiguous); -> (cos [nq, M, 10], meta [nq, M, 3]) of the
        first nq queries, the bitwise OR over the G parts (msr_rerank_combine)."""
        G, M = int(cos_parts.shape[0]), int(cos_parts.shape[2])
        stride = cos_parts.stride(0) * 4 if G > 1 else 0
        assert G == 1 or meta_parts.stride(0) * 4 == stride
        assert cos_parts[0].is_contiguous() and meta_parts[0].is_contiguous()
        cos = torch.empty((nq, M, _abi.MSR_RERANK_MAX_CHUNKS), dtype=torch.float32, device=self.device)
        meta = torch.empty((nq, M, 3), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_rerank_combine(self.handle, _ptr(cos_parts), _ptr(meta_parts), G, int(stride), int(nq), M,
                                                _ptr(cos), _ptr(meta), self._stream()))
        return cos, meta

    # -- the compact form of the rerank exchange (msretr.h: msr_rerank_plan / _gather_records / _scatter)
    def rerank_plan(self, cand_doc_global, cand_n, shard_bounds, my_shard, queries_per_shard, plan):
        """Fills `plan` (distributed._RerankPlan: counts [N, Q], send_base [Q], send_blk [Q, ceil(M / 8)], recv_off [N, Qs],
        pair [N, N], all int32 on the device) for the merged candidate lists cand_doc_global [Q, M] / cand_n [Q]."""
        cand = self._dev(cand_doc_global, torch.int32)
        cn = self._dev(cand_n, torch.int32)
        Q, M = int(cand.shape[0]), int(cand.shape[1])
        N = int(shard_bounds.numel()) - 1
        assert tuple(plan.counts.shape) == (N, Q) and tuple(plan.send_blk.shape) == (Q, (M + 7) // 8)
        assert tuple(plan.recv_off.shape) == (N, int(queries_per_shard)) and tuple(plan.pair.shape) == (N, N)
        self._check(self.lib.msr_rerank_plan(self.handle, Q, _ptr(cand), _ptr(cn), M, _ptr(shard_bounds), N, int(my_shard),
                                             int(queries_per_shard), _ptr(plan.counts), _ptr(plan.send_base), _ptr(plan.send_blk),
                                             _ptr(plan.recv_off), _ptr(plan.pair), self._stream()))

    def rerank_gather_records(self, qvec, cand_doc_global, cand_n, plan, records, doc_base=0, row_base=0, max_chunks=10):
        """rerank_gather for ALL queries in one launch, as 16-word records of the slots this shard owns, at the places `plan`
        holds (records: int32, at least sum(plan.pair[my]) * 16 words)."""
        q = self._dev(qvec, torch.float32).reshape(-1, DIM)
        cand = self._dev(cand_doc_global, torch.int32)
        cn = self._dev(cand_n, torch.int32)
        Q, M = int(cand.shape[0]), int(cand.shape[1])
        assert records.is_contiguous() and records.dtype == torch.int32
        self._check(self.lib.msr_rerank_gather_records(self.handle, _ptr(q), Q, _ptr(cand), _ptr(cn), M, int(doc_base), int(row_base),
                                                       int(max_chunks), _ptr(plan.send_base), _ptr(plan.send_blk), _ptr(records),
                                                       int(records.numel()) // 16, self._stream()))

    def rerank_scatter(self, records, plan, first_query, nq, M):
        """The received records of my queries [first_query, first_query + nq) -> (cos [nq, M, 10], meta [nq, M, 3])."""
        N, Q = int(plan.counts.shape[0]), int(plan.counts.shape[1])
        cos = torch.empty((nq, M, _abi.MSR_RERANK_MAX_CHUNKS), dtype=torch.float32, device=self.device)
        meta = torch.empty((nq, M, 3), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_rerank_scatter(self.handle, _ptr(records), int(records.numel()) // 16, _ptr(plan.counts), _ptr(plan.recv_off), N, Q,
                                                int(plan.recv_off.shape[1]), int(first_query), int(nq), int(M), _ptr(cos), _ptr(meta),
                                                self._stream()))
        return cos, meta

    def rerank_fuse(self, cand_doc_global, cand_bm25, cand_n, cos, meta, **params):
        p = dict(RERANK_DEFAULTS)
        p.update(params)
        cand = self._dev(cand_doc_global, torch.int32)
        bm = self._dev(cand_bm25, torch.float64)
        cn = self._dev(cand_n, torch.int32)
        Q, M = int(cand.shape[0]), int(cand.shape[1])
        prm = _abi.MsrRerankParams(p["smoothing"], p["max_boost"], p["max_decay"], int(p["max_chunks"]), 0)
        mk = lambda dt: torch.empty((Q, M), dtype=dt, device=self.device)
        out_doc, out_score, out_orig, out_chunk = mk(torch.int32), mk(torch.float64), mk(torch.float64), mk(torch.int32)
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        out_rows = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_rerank_fuse(self.handle, Q, _ptr(cand), _ptr(bm), _ptr(cn), M, _ptr(cos.contiguous()),
                                             _ptr(meta.contiguous()), C.byref(prm), _ptr(out_doc), _ptr(out_score),
                                             _ptr(out_orig), _ptr(out_chunk), _ptr(out_n), _ptr(out_rows),
                                             self._stream()))
        return out_doc, out_score, out_orig, out_chunk, out_n, out_rows

    # ------------------------------------------------------------------ response assembly on the device
    def bind_doc_domains(self, domain):
        """domain int32 [N_global]: domain id of every document (equal ids = same urlparse(url).netloc.lower()), -1 = a
        document the response models reject (NULL title / url / text, reranker_api.py:376-397).  None unbinds."""
        t = None if domain is None else self._dev(domain, torch.int32)
        self._t["doc_domain"] = t
        self._check(self.lib.msr_bind_doc_domains(self.handle, _ptr(t), 0 if t is None else int(t.numel()), self._stream()))

    def diversify(self, fused, top_k=100, relevance_threshold=0.8, diversification=True):
        """fused = (doc, score, orig, chunk, n, ...) of rerank / rerank_fuse -> (doc, score f64, orig f64, chunk, n): the final
        list of every query after reranker_api.py:178-236 (or, diversification=False, its first top_k accepted entries)."""
        doc, score, orig, chunk, n = fused[:5]
        Q, M = int(doc.shape[0]), int(doc.shape[1])
        mk = lambda dt: torch.empty((Q, M), dtype=dt, device=self.device)
        o_doc, o_score, o_orig, o_chunk = mk(torch.int32), mk(torch.float64), mk(torch.float64), mk(torch.int32)
        o_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_diversify(self.handle, Q, _ptr(doc.contiguous()), _ptr(score.contiguous()), _ptr(orig.contiguous()),
                                           _ptr(chunk.contiguous()), _ptr(n.contiguous()), M, int(top_k),
                                           C.c_double(relevance_threshold), 1 if diversification else 0, _ptr(o_doc),
                                           _ptr(o_score), _ptr(o_orig), _ptr(o_chunk), _ptr(o_n), self._stream()))
        return o_doc, o_score, o_orig, o_chunk, o_n

    # ------------------------------------------------------------------ shard merge
    def merge_topk(self, docs, scores, ns, k):
        """docs int32 [G, Q, k] GLOBAL indices, scores f32/f64 [G, Q, k], ns int32 [G, Q] -> merged top-k."""
        G, Q = int(docs.shape[0]), int(docs.shape[1])
        bits = 64 if scores.dtype == torch.float64 else 32
        out_doc = torch.empty((Q, k), dtype=torch.int32, device=self.device)
        out_score = torch.empty((Q, k), dtype=scores.dtype, device=self.device)
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        self._check(self.lib.msr_merge_topk(self.handle, _ptr(docs.contiguous()), _ptr(scores.contiguous()),
                                            _ptr(ns.contiguous()), G, Q, k, bits, _ptr(out_doc), _ptr(out_score),
                                            _ptr(out_n), self._stream()))
        return out_doc, out_score, out_n

    def merge_gathered(self, ex, doc, score, n, payload, k):
        """Merge the `world` records of an all-gather IN PLACE (distributed._Exchange): the kernel reads segment `doc` /
        `score` / `n` (/ `payload`) of every rank's record through a byte stride.  -> (doc, score, n, payload | None)."""
        sc0 = ex.part(0, score)
        Q = ex.Q
        bits = 64 if sc0.dtype == torch.float64 else 32
        out_doc = torch.empty((Q, k), dtype=torch.int32, device=self.device)
        out_score = torch.empty((Q, k), dtype=sc0.dtype, device=self.device)
        out_n = torch.empty((Q,), dtype=torch.int32, device=self.device)
        out_pay = torch.empty((Q, k), dtype=torch.int32, device=self.device) if payload else None
        self._check(self.lib.msr_merge_topk_payload(
            self.handle, _ptr(ex.part(0, doc)), _ptr(sc0), _ptr(ex.part(0, n)), _ptr(ex.part(0, payload)) if payload else _ptr(None),
            ex.world, ex.record, Q, k, bits, _ptr(out_doc), _ptr(out_score), _ptr(out_n), _ptr(out_pay), self._stream()))
        return out_doc, out_score, out_n, out_pay

    # ------------------------------------------------------------------ timing hooks (bench.py)
    def set_timing(self, on):
        self._check(self.lib.msr_set_timing(self.handle, 1 if on else 0))

    def kernel_time_ms(self, which):
        ms, n = C.c_float(), C.c_int32()
        self._check(self.lib.msr_kernel_time_ms(self.handle, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value
