"""CPU stand-in with the DeviceEngine method surface, backed by the oracle.  Used ONLY by tests to drive
msretr.distributed.ShardedEngine under gloo (the exchange logic has no GPU dependency; the kernels do)."""
import numpy as np
import torch

from oracle import bm25_ref, dense_ref, rerank_ref


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class OracleEngine:
    def __init__(self, index):
        self.index = index
        self.z = {k: _np(getattr(index, k)) for k in ("doc_ids", "doc_len", "term_off", "post_doc", "post_tf", "idf")}
        self.z["avgdl"] = index.avgdl
        self.doc_off = _np(index.doc_off).astype(np.int64)
        self.emb = _np(index.emb)
        self.url_group = np.asarray(index.url_group())

    def bm25_topk(self, term_lists, k=1000, min_score=0.0, packed=None):
        Q = len(term_lists)
        doc = np.full((Q, k), -1, np.int32); sc = np.full((Q, k), -np.inf, np.float64); n = np.zeros(Q, np.int32)
        for i, t in enumerate(term_lists):
            a, b = bm25_ref.topk(self.z, t, k, min_score, self.index.k1, self.index.b)
            doc[i, :len(a)], sc[i, :len(a)], n[i] = a, b, len(a)
        return torch.as_tensor(doc), torch.as_tensor(sc), torch.as_tensor(n)

    def dense_topk(self, qvec, k=100, max_chunks_per_doc=0, want_chunk=True):
        q = _np(qvec)
        Q = len(q)
        doc = np.full((Q, k), -1, np.int32); sc = np.full((Q, k), -np.inf, np.float32)
        ch = np.full((Q, k), -1, np.int32); n = np.zeros(Q, np.int32)
        for i in range(Q):
            a, b, c = dense_ref.quick_search(self.emb, self.doc_off, q[i], k, max_chunks_per_doc)
            doc[i, :len(a)], sc[i, :len(a)], ch[i, :len(a)], n[i] = a, b, c, len(a)
        return torch.as_tensor(doc), torch.as_tensor(sc), torch.as_tensor(ch), torch.as_tensor(n)

    # the two halves of the dense call (DeviceEngine.dense_begin / dense_end): the stand-in vouches for the EXACT score of its
    # k_part-th best document and afterwards returns only the documents at or above the bound the shards agreed on
    dense_split_min = 1

    def dense_split_max(self, k=100):
        return 1 << 20

    def dense_begin(self, qvec, k=100, k_part=None):
        self.begin_calls = getattr(self, "begin_calls", 0) + 1
        self._pending = self.dense_topk(qvec, k=k)
        _, sc, _, n = [_np(x) for x in self._pending]
        kp = int(k_part or k)
        part = np.where(n >= kp, sc[:, kp - 1], -np.inf).astype(np.float32)
        return torch.as_tensor(part)

    def dense_end(self, Q, k=100, bound=None, want_chunk=True, out=None):
        doc, sc, ch, n = [_np(x).copy() for x in self._pending]
        assert len(doc) == Q
        if bound is not None:
            b = _np(bound)
            for q in range(Q):
                keep = int((sc[q, :n[q]] >= b[q]).sum())           # (the list is sorted: a prefix)
                doc[q, keep:], sc[q, keep:], ch[q, keep:], n[q] = -1, -np.inf, -1, keep
        res = torch.as_tensor(doc), torch.as_tensor(sc), torch.as_tensor(ch), torch.as_tensor(n)
        if out is not None:
            for dst, src in zip(out, res):
                dst.copy_(src)
            return out
        return res

    def merge_topk(self, docs, scores, ns, k):
        docs, scores, ns = _np(docs), _np(scores), _np(ns)
        G, Q = docs.shape[:2]
        od = np.full((Q, k), -1, np.int32); os_ = np.full((Q, k), -np.inf, scores.dtype); on = np.zeros(Q, np.int32)
        for q in range(Q):
            a, b = dense_ref.merge_topk([(docs[g, q, :ns[g, q]].astype(np.int64), scores[g, q, :ns[g, q]]) for g in range(G)], k)
            od[q, :len(a)], os_[q, :len(a)], on[q] = a, b, len(a)
        return torch.as_tensor(od), torch.as_tensor(os_), torch.as_tensor(on)

    def merge_gathered(self, ex, doc, score, n, payload, k):
        """Stand-in for DeviceEngine.merge_gathered: same in-place reading of the gathered records, merge by the oracle,
        payload looked up from the merged (doc, part) -- documents are owned by exactly one shard."""
        G, Q = ex.world, ex.Q
        docs = np.stack([_np(ex.part(g, doc)) for g in range(G)])
        scores = np.stack([_np(ex.part(g, score)) for g in range(G)])
        ns = np.stack([_np(ex.part(g, n)) for g in range(G)])
        od, os_, on = [_np(x) for x in self.merge_topk(docs, scores, ns, k)]
        op = None
        if payload:
            pays = np.stack([_np(ex.part(g, payload)) for g in range(G)])
            op = np.full((Q, k), -1, np.int32)
            for q in range(Q):
                look = {int(docs[g, q, r]): int(pays[g, q, r]) for g in range(G) for r in range(ns[g, q])}
                for r in range(on[q]):
                    op[q, r] = look[int(od[q, r])]
            op = torch.as_tensor(op)
        return torch.as_tensor(od), torch.as_tensor(os_), torch.as_tensor(on), op

    def rerank_combine(self, cos_parts, meta_parts, nq):
        c = np.bitwise_or.reduce(_np(cos_parts).view(np.uint32), axis=0)[:nq]
        m = np.bitwise_or.reduce(_np(meta_parts), axis=0)[:nq]
        return torch.as_tensor(np.ascontiguousarray(c).view(np.float32)), torch.as_tensor(np.ascontiguousarray(m))

    def rerank_gather(self, qvec, cand_doc_global, cand_n, doc_base=0, row_base=0, max_chunks=10, out=None):
        q, cand, cn = _np(qvec), _np(cand_doc_global), _np(cand_n)
        Q, M = cand.shape
        cos = np.zeros((Q, M, 10), np.float32); meta = np.zeros((Q, M, 3), np.int32)
        N = len(self.doc_off) - 1
        for i in range(Q):
            for m in range(cn[i]):
                d = cand[i, m] - doc_base
                if d < 0 or d >= N:
                    continue
                lo, hi = self.doc_off[d], min(self.doc_off[d + 1], self.doc_off[d] + max_chunks)
                if hi > lo:
                    cos[i, m, :hi - lo] = rerank_ref.cosine_f32(q[i], self.emb[lo:hi])
                meta[i, m] = (hi - lo, self.url_group[d] + 2, lo + row_base)
        if out is not None:
            out[0].copy_(torch.as_tensor(cos)); out[1].copy_(torch.as_tensor(meta))
            return out
        return torch.as_tensor(cos), torch.as_tensor(meta)

    def rerank_gather_blocks(self, qvec, cand_doc_global, cand_n, blocks, queries_per_block, doc_base=0, row_base=0, max_chunks=10):
        """Stand-in for DeviceEngine.rerank_gather_blocks: block b = [cos of queries b * qpb .. | their meta | padding]."""
        cos, meta = self.rerank_gather(qvec, cand_doc_global, cand_n, doc_base, row_base, max_chunks)
        Q, M = cos.shape[0], cos.shape[1]
        qpb = queries_per_block
        for b in range(blocks.shape[0]):
            lo, hi = min(Q, b * qpb), min(Q, (b + 1) * qpb)
            if hi > lo:
                blocks[b, :(hi - lo) * M * 10].copy_(cos[lo:hi].reshape(-1).view(torch.int32))
                blocks[b, qpb * M * 10: qpb * M * 10 + (hi - lo) * M * 3].copy_(meta[lo:hi].reshape(-1))

    # the compact exchange (DeviceEngine.rerank_plan / rerank_gather_records / rerank_scatter; msretr.h)
    def rerank_plan(self, cand_doc_global, cand_n, shard_bounds, my_shard, queries_per_shard, plan):
        cand, cn, bounds = _np(cand_doc_global).astype(np.int64), _np(cand_n), _np(shard_bounds).astype(np.int64)
        Q, M = cand.shape
        N, qps = len(bounds) - 1, int(queries_per_shard)
        valid = (np.arange(M)[None, :] < cn[:, None]) & (cand >= bounds[0]) & (cand < bounds[-1])
        owner = np.where(valid, np.searchsorted(bounds, cand, side="right") - 1, -1)
        counts = np.stack([(owner == s).sum(axis=1) for s in range(N)]).astype(np.int32)
        mine = np.zeros((Q, (M + 7) // 8 * 8), np.int64)
        mine[:, :M] = owner == my_shard
        per_blk = mine.reshape(Q, -1, 8).sum(axis=2)
        lo, hi = min(Q, my_shard * qps), min(Q, (my_shard + 1) * qps)
        pair = np.array([[counts[s, min(Q, o * qps):min(Q, (o + 1) * qps)].sum() for o in range(N)] for s in range(N)], np.int32)
        recv_off = np.zeros((N, qps), np.int32)
        base = 0
        for s in range(N):
            c = counts[s, lo:hi]
            recv_off[s, :hi - lo] = base + np.cumsum(c) - c
            base += int(c.sum())
        plan.counts.copy_(torch.as_tensor(counts))
        plan.send_base.copy_(torch.as_tensor((np.cumsum(counts[my_shard]) - counts[my_shard]).astype(np.int32)))
        plan.send_blk.copy_(torch.as_tensor((np.cumsum(per_blk, axis=1) - per_blk).astype(np.int32)))
        plan.recv_off.copy_(torch.as_tensor(recv_off))
        plan.pair.copy_(torch.as_tensor(pair))

    def rerank_gather_records(self, qvec, cand_doc_global, cand_n, plan, records, doc_base=0, row_base=0, max_chunks=10):
        cos, meta = [_np(x) for x in self.rerank_gather(qvec, cand_doc_global, cand_n, doc_base, row_base, max_chunks)]
        cand, cn = _np(cand_doc_global), _np(cand_n)
        Q, M = cand.shape
        N = len(self.doc_off) - 1
        rec = _np(records).reshape(-1, 16)
        base = _np(plan.send_base)
        for q in range(Q):
            r = int(base[q])
            for m in range(min(int(cn[q]), M)):
                if 0 <= cand[q, m] - doc_base < N:
                    rec[r, 0], rec[r, 1:4], rec[r, 14], rec[r, 15] = m, meta[q, m], q, 0
                    rec[r, 4:14] = cos[q, m].view(np.int32)
                    r += 1
        records.copy_(torch.as_tensor(rec.reshape(-1)))

    def rerank_scatter(self, records, plan, first_query, nq, M):
        rec, counts, off = _np(records).reshape(-1, 16), _np(plan.counts), _np(plan.recv_off)
        cos = np.zeros((nq, M, 10), np.float32); meta = np.zeros((nq, M, 3), np.int32)
        for s in range(counts.shape[0]):
            for j in range(nq):
                for r in rec[off[s, j]: off[s, j] + counts[s, first_query + j]]:
                    assert r[14] == first_query + j, "a record of another query"
                    meta[j, r[0]] = r[1:4]
                    cos[j, r[0]] = r[4:14].view(np.float32)
        return torch.as_tensor(cos), torch.as_tensor(meta)

    def rerank_fuse(self, cand_doc_global, cand_bm25, cand_n, cos, meta, smoothing=0.15, max_boost=0.1,
                    max_decay=0.05, max_chunks=10):
        cand, bm, cn, cos, meta = map(_np, (cand_doc_global, cand_bm25, cand_n, cos, meta))
        Q, M = cand.shape
        out_doc = np.full((Q, M), -1, np.int32); out_score = np.full((Q, M), -np.inf); out_orig = np.zeros((Q, M))
        out_chunk = np.full((Q, M), -1, np.int32); out_n = np.zeros(Q, np.int32); out_rows = np.zeros(Q, np.int32)
        for i in range(Q):
            best_of_group = {}
            for m in range(cn[i]):
                g = meta[i, m, 1] - 2
                if cand[i, m] >= 0 and g >= 0:
                    if g not in best_of_group or cand[i, m] < cand[i, best_of_group[g]]:
                        best_of_group[g] = m
            kept = sorted((m for m in best_of_group.values() if meta[i, m, 0] > 0), key=lambda m: cand[i, m])
            if not kept:
                continue
            flat = [(m, j) for m in kept for j in range(meta[i, m, 0])]
            new = rerank_ref.normalise([float(cos[i, m, j]) for m, j in flat])
            old = rerank_ref.normalise([float(bm[i, m]) for m, _ in flat])
            new = [a * (1 - smoothing) + b * smoothing for a, b in zip(new, old)]
            res, p = [], 0
            for m in kept:
                n = int(meta[i, m, 0])
                adj = rerank_ref.positional_adjust(new[p:p + n], n)
                b = max(range(n), key=lambda t: (adj[t], -t))
                res.append((-adj[b], int(cand[i, m]), adj[b], old[p + b], int(meta[i, m, 2]) + b))
                p += n
            res.sort()
            for r, (_, d, s, o, c) in enumerate(res):
                out_doc[i, r], out_score[i, r], out_orig[i, r], out_chunk[i, r] = d, s, o, c
            out_n[i], out_rows[i] = len(res), len(flat)
        return tuple(torch.as_tensor(x) for x in (out_doc, out_score, out_orig, out_chunk, out_n, out_rows))
