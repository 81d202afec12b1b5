"""Doc-sharded execution: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The corpus is cut into contiguous ranges of the dense document index (balanced by chunk count,
CorpusIndex.shard).  Global statistics (idf, avgdl) are replicated, so a shard's BM25 scores are bit-equal
to the unsharded ones.  Per query batch:

  1. every rank: BM25 top-k1 and dense top-k2 over its shard                 (msr_bm25_topk / msr_dense_topk)
  2. ONE all-gather of the packed per-shard lists (k1*(4+8) + k2*(4+4+4) bytes per query)
  3. every rank: the same deterministic merge (score desc, doc index asc)    (msr_merge_topk)
  4. reference-exact rerank of the GLOBAL stage-1 candidates: each rank computes the cosines of the
     candidates it owns (msr_rerank_gather), one all-reduce (integer SUM over the raw bits: exactly one rank
     contributes non-zero bits per candidate, so the sum is a select), and every rank runs the float64 chain (msr_rerank_fuse).

No embedding or posting ever crosses a link: only k records per query do.  The reference has no counterpart
(it is a single process talking HTTP to itself).
"""
import torch
import torch.distributed as dist


class ShardedEngine:
    """`engine` is a DeviceEngine bound to this rank's shard (or any object with the same five methods:
    bm25_topk, dense_topk, merge_topk, rerank_gather, rerank_fuse -- the CPU tests pass an oracle-backed
    stand-in to exercise the exchange logic under gloo)."""

    def __init__(self, engine, doc_base, row_base, group=None):
        self.engine = engine
        self.doc_base = int(doc_base)
        self.row_base = int(row_base)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0

    # ------------------------------------------------------------------ exchange helpers
    def _allgather_bytes(self, parts):
        """parts: list of tensors -> list (per rank) of lists of tensors with the same shapes/dtypes.
        One collective: everything is packed into a single byte buffer (8-byte aligned segments, so the
        float64 scores can be viewed in place on the receiving side)."""
        segs, sizes = [], []
        for p in parts:
            b = p.contiguous().view(torch.uint8).reshape(-1)
            pad = (-b.numel()) % 8
            if pad:
                b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
            segs.append(b)
            sizes.append(b.numel())
        flat = torch.cat(segs)
        out = torch.empty(self.world * flat.numel(), dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(out, flat, group=self.group)
        out = out.view(self.world, flat.numel())
        res = []
        for g in range(self.world):
            o, lst = 0, []
            for p, sz in zip(parts, sizes):
                nb = p.numel() * p.element_size()
                lst.append(out[g, o:o + nb].view(p.dtype).reshape(p.shape))
                o += sz
            res.append(lst)
        return res

    @staticmethod
    def _globalise(idx, base):
        return idx if base == 0 else torch.where(idx >= 0, idx + base, idx)

    # ------------------------------------------------------------------ the sharded hot path
    def search(self, term_lists, qvec, k1=1000, k2=100, min_score=0.0, max_chunks_per_doc=0, rerank=True,
               packed=None, dense_batched=False, **rerank_params):
        e = self.engine
        b_doc, b_score, b_n = e.bm25_topk(term_lists, k=k1, min_score=min_score, packed=packed)
        dense = e.dense_topk_batched if dense_batched else e.dense_topk
        d_doc, d_score, d_chunk, d_n = dense(qvec, k=k2, max_chunks_per_doc=max_chunks_per_doc)
        b_doc = self._globalise(b_doc, self.doc_base)
        d_doc = self._globalise(d_doc, self.doc_base)
        d_chunk = self._globalise(d_chunk, self.row_base)
        if self.world > 1:
            parts = self._allgather_bytes([b_doc, b_score, b_n, d_doc, d_score, d_chunk, d_n])
            stack = lambda j: torch.stack([p[j] for p in parts])
            b_doc, b_score, b_n = e.merge_topk(stack(0), stack(1), stack(2), k1)
            g_doc, g_score, g_chunk, g_n = stack(3), stack(4), stack(5), stack(6)
            d_doc, d_score, d_n = e.merge_topk(g_doc, g_score, g_n, k2)
            # the arg-max chunk travels with its document: look it up among the gathered lists
            flat_doc = g_doc.permute(1, 0, 2).reshape(g_doc.shape[1], -1)
            flat_chunk = g_chunk.permute(1, 0, 2).reshape(g_doc.shape[1], -1)
            hit = (flat_doc.unsqueeze(1) == d_doc.unsqueeze(2)) & (d_doc.unsqueeze(2) >= 0)
            pos = hit.to(torch.int8).argmax(dim=2)
            d_chunk = torch.where(d_doc >= 0, torch.gather(flat_chunk, 1, pos), torch.full_like(d_doc, -1))
        out = dict(bm25=(b_doc, b_score, b_n), dense=(d_doc, d_score, d_chunk, d_n))
        if rerank:
            cos, meta = e.rerank_gather(qvec, b_doc, b_n, doc_base=self.doc_base, row_base=self.row_base,
                                        max_chunks=rerank_params.get("max_chunks", 10))
            if self.world > 1:
                buf = torch.cat([cos.view(torch.int32).reshape(-1), meta.reshape(-1)])
                # integer SUM of the raw bits: exactly one rank holds non-zero bits per word, so the sum IS that word
                # (RCCL/NCCL has no bitwise reduce op; float SUM would also be exact here but -0.0 + 0.0 is not)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                cos = buf[:cos.numel()].view(torch.float32).reshape(cos.shape)
                meta = buf[cos.numel():].reshape(meta.shape)
            out["rerank"] = e.rerank_fuse(b_doc, b_score, b_n, cos, meta, **rerank_params)
        return out
