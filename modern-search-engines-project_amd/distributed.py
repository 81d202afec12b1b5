"""Doc-sharded execution: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The corpus is cut into contiguous ranges of the dense document index (balanced by chunk count,
CorpusIndex.shard).  Global statistics (idf, avgdl) are replicated, so a shard's BM25 scores are bit-equal
to the unsharded ones.  Per query batch:

  1. every rank: BM25 top-k1 over its shard (msr_bm25_topk); ONE all-gather of the packed lists (k1 * (4 + 8) bytes per
     query; send and receive buffers allocated once per batch shape); every rank: the same deterministic merge (score desc,
     doc index asc), reading the gathered records in place (msr_merge_topk_payload).  The merged candidate lists are
     replicated and the shards are document ranges, so every rank can now COUNT who owns which candidate
     (msr_rerank_plan); the N x N matrix of counts goes to the host beside step 2.
  2. every rank: dense top-k2 over its shard, the shards agreeing on a bound first (one all-reduce MIN of a float per
     query, see _dense); ONE all-gather of the lists (k2 * 12 bytes per query), merge; the arg-max chunk row of a dense
     entry rides along as the merge payload.
  3. reference-exact rerank of the GLOBAL stage-1 candidates, sharded BY QUERY for everything that is not tied to the
     documents: rank r owns queries [r Qs, (r + 1) Qs), Qs = ceil(Q / world).
       a. every rank computes the cosines / meta of the candidates it owns, for all queries, as 16-word RECORDS of the owned
          slots only (msr_rerank_gather_records), straight into the send buffer of
       b. ONE all-to-all with the split sizes of step 1's matrix (the host waits for that copy here -- it was made before
          the dense stage ran, so the wait is over before it starts): the records of query q go to the rank that owns q
          (xGMI is point to point: every link carries its pair's records at the same time).  1 / world of the dense form's
          bytes (52 KB per query and rank at k1 = 1000, mostly zero words; `a2a="blocks"` keeps that form:
          msr_rerank_gather_blocks + msr_rerank_combine);
       c. the owner scatters the records into the dense arrays (msr_rerank_scatter) and runs the float64 chain for ITS
          queries only (msr_rerank_fuse: 1 / world of the work);
       d. ONE all-gather of the fused lists (rerank_keep entries per query) gives every rank the result.

No embedding or posting ever crosses a link: only k records per query do.  The reference has no counterpart
(it is a single process talking HTTP to itself).
"""
import torch
import torch.distributed as dist


_SEGS_B = (("b_doc", torch.int32, "k1"), ("b_score", torch.float64, "k1"), ("b_n", torch.int32, None))
_SEGS_D = (("d_doc", torch.int32, "k2"), ("d_score", torch.float32, "k2"), ("d_chunk", torch.int32, "k2"),
           ("d_n", torch.int32, None))
_SEGS = _SEGS_B + _SEGS_D


class _Exchange:
    """The record one rank contributes to an all-gather -- [b_doc | b_score | b_n] after stage 1, [d_doc | d_score | d_chunk |
    d_n] after stage 2, every segment 8-byte aligned -- and the receive buffer [world][record], allocated ONCE per (Q, k1,
    k2, device).  The merge kernels read the gathered segments in place (msr_merge_topk_payload with part_stride_bytes =
    len(record))."""

    def __init__(self, world, Q, k1, k2, device, segs=_SEGS):
        self.off, o = {}, 0
        for name, dt, kk in segs:
            n = Q * (k1 if kk == "k1" else k2 if kk == "k2" else 1)
            self.off[name] = (o, n, dt)
            o += (n * torch.empty(0, dtype=dt).element_size() + 7) // 8 * 8
        self.record = o
        self.world, self.Q, self.k1, self.k2 = world, Q, k1, k2
        self.send = torch.zeros(o, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(world * o, dtype=torch.uint8, device=device) if world > 1 else self.send
        self.rerank = {}                                  # (M, keep) -> _RerankExchange

    def _view(self, buf, base, name):
        o, n, dt = self.off[name]
        es = torch.empty(0, dtype=dt).element_size()
        v = buf[base + o: base + o + n * es].view(dt)
        kk = dict((a, c) for a, _, c in _SEGS)[name]
        return v.view(self.Q, self.k1 if kk == "k1" else self.k2) if kk else v

    def out(self, name):                                  # this rank's segment (the kernels write their results here)
        return self._view(self.send, 0, name)

    def part(self, g, name):                              # rank g's segment of the gathered buffer
        return self._view(self.recv, g * self.record, name)


_FUSED = (("doc", torch.int32, True), ("score", torch.float64, True), ("orig", torch.float64, True),
          ("chunk", torch.int32, True), ("n", torch.int32, False), ("rows", torch.int32, False))


RECORD_WORDS = 16          # msretr.h: [slot, rows, url group + 2, first row, cos x 10, query, 0]


class _RerankPlan:
    """What msr_rerank_plan fills (device, int32) + the host copy of the N x N matrix the all-to-all is sized with."""

    def __init__(self, world, Q, Qs, M, device):
        z = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=device)
        self.counts, self.send_base, self.send_blk = z(world, Q), z(Q), z(Q, (M + 7) // 8)
        self.recv_off, self.pair = z(world, Qs), z(world, world)
        cuda = torch.device(device).type == "cuda"
        self.pair_host = torch.zeros((world, world), dtype=torch.int32, pin_memory=cuda)
        self.event = torch.cuda.Event() if cuda else None

    def to_host(self):
        """Enqueue the copy of `pair` to the host (nothing waits here)."""
        self.pair_host.copy_(self.pair, non_blocking=True)
        if self.event is not None:
            self.event.record()

    def splits(self, rank):
        """-> (send split sizes, receive split sizes) in 32-bit words; waits for the copy enqueued by to_host."""
        if self.event is not None:
            self.event.synchronize()
        m = self.pair_host.tolist()
        return [m[rank][o] * RECORD_WORDS for o in range(len(m))], [m[g][rank] * RECORD_WORDS for g in range(len(m))]


class _RerankExchange:
    """Buffers of the query-sharded rerank, allocated once per (Q, M, keep, device).
    a2a_send / a2a_recv: int32 [world][block]; block o of the send buffer holds this rank's halves of the queries rank o owns,
    cos [Qs][M][10] (float bits) followed by meta [Qs][M][3]; block g of the receive buffer is rank g's half of MY queries.
    out_send / out_recv: the fused lists of my queries, [doc | score | orig | chunk | n | rows] with `keep` entries per query,
    every segment 8-byte aligned, and the gathered records of all ranks."""

    def __init__(self, world, Q, M, keep, device):
        self.world, self.Q, self.M, self.keep = world, Q, M, keep
        self.Qs = (Q + world - 1) // world
        self.cos_words = self.Qs * M * 10
        self.block = (self.Qs * M * 13 + 3) // 4 * 4      # whole 16-byte units: the join takes its wide path
        self.a2a_send = torch.zeros(world * self.block, dtype=torch.int32, device=device)
        self.a2a_recv = torch.zeros(world * self.block, dtype=torch.int32, device=device)
        self.off, o = {}, 0
        for name, dt, per_k in _FUSED:
            n = self.Qs * (keep if per_k else 1)
            self.off[name] = (o, n, dt)
            o += (n * torch.empty(0, dtype=dt).element_size() + 7) // 8 * 8
        self.record = o
        self.out_send = torch.zeros(o, dtype=torch.uint8, device=device)
        self.out_recv = torch.zeros(world * o, dtype=torch.uint8, device=device)
        self.device = device
        self.plan = self.rec_send = self.rec_recv = None

    def records(self):
        """The buffers of the compact form, allocated at first use for the worst case: every candidate of every query in THIS
        shard on the way out (Q x M records), every slot of my queries owned by somebody on the way in (Qs x M)."""
        if self.plan is None:
            self.plan = _RerankPlan(self.world, self.Q, self.Qs, self.M, self.device)
            self.rec_send = torch.zeros(self.Q * self.M * RECORD_WORDS, dtype=torch.int32, device=self.device)
            self.rec_recv = torch.zeros(self.Qs * self.M * RECORD_WORDS, dtype=torch.int32, device=self.device)
        return self.plan

    def send_views(self, o):
        """(cos float32 [Qs, M, 10], meta int32 [Qs, M, 3]) of block o of the send buffer."""
        b = self.a2a_send[o * self.block:(o + 1) * self.block]
        return (b[:self.cos_words].view(torch.float32).view(self.Qs, self.M, 10),
                b[self.cos_words:self.Qs * self.M * 13].view(self.Qs, self.M, 3))

    def recv_parts(self):
        """(cos float32 [world, Qs, M, 10], meta int32 [world, Qs, M, 3]): strided views of the receive buffer."""
        r = self.a2a_recv.view(self.world, self.block)
        return (r[:, :self.cos_words].view(torch.float32).view(self.world, self.Qs, self.M, 10),
                r[:, self.cos_words:self.Qs * self.M * 13].view(self.world, self.Qs, self.M, 3))

    def out_view(self, name):
        o, n, dt = self.off[name]
        v = self.out_send[o:o + n * torch.empty(0, dtype=dt).element_size()].view(dt)
        return v.view(self.Qs, self.keep) if dict((a, c) for a, _, c in _FUSED)[name] else v

    def gathered(self, name):
        """Field `name` of all ranks in query order: [world * Qs, keep] (or [world * Qs]); rows >= Q are padding."""
        o, n, dt = self.off[name]
        es = torch.empty(0, dtype=dt).element_size()
        v = self.out_recv.view(self.world, self.record)[:, o:o + n * es].contiguous().view(dt)
        return v.view(self.world * self.Qs, self.keep) if dict((a, c) for a, _, c in _FUSED)[name] else v.view(-1)


class ShardedEngine:
    """`engine` is a DeviceEngine bound to this rank's shard (or any object with the same methods: bm25_topk, dense_topk,
    merge_gathered, rerank_gather, rerank_fuse -- the CPU tests pass an oracle-backed stand-in to exercise the exchange
    logic under gloo)."""

    def __init__(self, engine, doc_base, row_base, group=None, a2a="records"):
        """a2a: "records" (the compact rerank exchange; needs an engine with rerank_plan and shards that are consecutive
        document ranges in rank order -- otherwise "blocks" is used) or "blocks" (dense halves)."""
        self.engine = engine
        self.doc_base = int(doc_base)
        self.row_base = int(row_base)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self._ex = {}
        assert a2a in ("records", "blocks")
        self.a2a = a2a
        self._bounds = None

    # ------------------------------------------------------------------ exchange helpers
    def _exchange(self, Q, k1, k2, device):
        """-> (stage-1 exchange, stage-2 exchange) for this batch shape."""
        key = (Q, k1, k2, str(device), self.world)
        if key not in self._ex:
            self._ex[key] = (_Exchange(self.world, Q, k1, k2, device, _SEGS_B), _Exchange(self.world, Q, k1, k2, device, _SEGS_D))
        return self._ex[key]

    def _shard_bounds(self, device):
        """Device int32 [world + 1]: shard g owns global documents bounds[g] <= doc < bounds[g + 1] -- ONE small all-gather,
        once.  None when the shards are not consecutive ranges in rank order, or the engine cannot tell its document count
        (then the rerank exchange keeps its dense form)."""
        if self._bounds is None:
            n = getattr(getattr(self.engine, "index", None), "n_docs", None)
            mine = torch.tensor([self.doc_base, self.doc_base + int(n) if n is not None else -1], dtype=torch.int64, device=device)
            every = torch.empty(self.world * 2, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            every = every.view(self.world, 2).cpu()
            ok = bool((every[:, 1] >= every[:, 0]).all()) and bool((every[1:, 0] == every[:-1, 1]).all())
            b = torch.cat([every[:, 0], every[-1:, 1]]).to(torch.int32).to(device) if ok and int(every[-1, 1]) < 2 ** 31 else None
            self._bounds = (b,)
        return self._bounds[0]

    def _allgather_bytes(self, parts):
        """parts: list of tensors -> list (per rank) of lists of tensors with the same shapes/dtypes.  One collective:
        everything is packed into a single byte buffer (8-byte aligned segments).  General-purpose helper (bench.py's
        single-stage workloads, tests); the hot path (search) uses the preallocated _Exchange instead."""
        sizes = [(p.numel() * p.element_size() + 7) // 8 * 8 for p in parts]
        flat = torch.zeros(sum(sizes), dtype=torch.uint8, device=parts[0].device)
        o = 0
        for p, sz in zip(parts, sizes):
            nb = p.numel() * p.element_size()
            flat[o:o + nb].copy_(p.contiguous().view(torch.uint8).reshape(-1))
            o += sz
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

    def _dense(self, qvec, k, max_chunks_per_doc, dense_batched, beside=None):
        """This shard's dense top-k.  With several shards and an engine that can split the call (DeviceEngine.dense_begin /
        dense_end) the shards first agree on a lower bound of the k-th cosine of the WHOLE corpus -- every shard vouches for
        ceil(k / world) of its own documents, ONE all-reduce MIN of a float per query -- and each then rescores in exact f32
        only what can be in the global top-k: 1 / world of the rescoring a rank otherwise does for its own top-k, for nothing.
        The lists come back shorter than k; merged they are the unsharded list, bit for bit.
        beside: called once, right after the stage's long pass over the shard has been enqueued (before the first collective
        of this stage) -- search() finishes stage 1's exchange there, so that its all-gather runs beside that pass."""
        e = self.engine
        beside = beside or (lambda: None)
        Q = int(qvec.shape[0]) if hasattr(qvec, "shape") else len(qvec)
        split = 0
        if self.world > 1 and not dense_batched and max_chunks_per_doc == 0 and hasattr(e, "dense_split_max"):
            split = e.dense_split_max(k)
        min_q = getattr(e, "dense_split_min", 65)             # (the device engine splits calls of more than 64 queries)
        if split <= 0 or Q < min_q:
            dense = e.dense_topk_batched if dense_batched else e.dense_topk
            res = dense(qvec, k=k, max_chunks_per_doc=max_chunks_per_doc)
            beside()
            return res
        k_part = (k + self.world - 1) // self.world
        if Q <= split:
            part = e.dense_begin(qvec, k=k, k_part=k_part)
            beside()
            dist.all_reduce(part, op=dist.ReduceOp.MIN, group=self.group)
            return e.dense_end(Q, k=k, bound=part)
        dev = getattr(e, "device", None) or (qvec.device if torch.is_tensor(qvec) else None)
        res = (torch.empty((Q, k), dtype=torch.int32, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev),
               torch.empty((Q, k), dtype=torch.int32, device=dev), torch.empty((Q,), dtype=torch.int32, device=dev))
        for a in range(0, Q, split):                          # pieces of the call write straight into their rows of the result
            b = min(Q, a + split)
            if b - a < min_q:                                 # (a short last piece: the sweeps, no split)
                for dst, src in zip(res, e.dense_topk(qvec[a:b], k=k)):
                    dst[a:b].copy_(src)
                continue
            part = e.dense_begin(qvec[a:b], k=k, k_part=k_part)
            if a == 0:
                beside()
            dist.all_reduce(part, op=dist.ReduceOp.MIN, group=self.group)
            e.dense_end(b - a, k=k, bound=part, out=tuple(t[a:b] for t in res))
        return res

    @staticmethod
    def _truncate(fused, keep):
        """rerank_keep: the first `keep` columns of the fused lists, with the COUNT clamped to match (a caller iterating
        range(n[q]) stays inside the truncated rows); `rows` (chunk rows that took part, RerankResponse.total_documents) is
        a property of the request, not of the list, and stays."""
        doc, score, orig, chunk, n, rows = fused
        return (doc[:, :keep], score[:, :keep], orig[:, :keep], chunk[:, :keep], torch.clamp(n, max=keep), rows)

    @staticmethod
    def _globalise(idx, base):
        return idx if base == 0 else torch.where(idx >= 0, idx + base, idx)

    @staticmethod
    def _globalise_into(dst, idx, base):
        if base == 0:
            dst.copy_(idx)
        else:
            torch.where(idx >= 0, idx + base, idx, out=dst)

    # ------------------------------------------------------------------ the sharded hot path
    def search(self, term_lists, qvec, k1=1000, k2=100, min_score=0.0, max_chunks_per_doc=0, rerank=True,
               packed=None, dense_batched=False, rerank_keep=None, **rerank_params):
        """-> dict(bm25=(doc, score, n), dense=(doc, score, chunk, n), rerank=(doc, score, orig, chunk, n, rows)); documents
        and chunk rows are GLOBAL indices; every rank returns the same tensors.  rerank_keep: entries per query of the fused
        lists to return (None: all k1; the reranker facade's diversification wants them all, a top-100 service k2)."""
        e = self.engine
        keep = k1 if rerank_keep is None else min(int(rerank_keep), k1)
        max_chunks = rerank_params.get("max_chunks", 10)
        b_doc, b_score, b_n = e.bm25_topk(term_lists, k=k1, min_score=min_score, packed=packed)
        Q = int(b_doc.shape[0])
        rx = plan = None
        if self.world > 1:
            ex, ex_d = self._exchange(Q, k1, k2, b_doc.device)
            # stage 1: this rank's record (local indices -> global, written straight into the preallocated send buffer), ONE
            # all-gather, the identical deterministic merge on every rank, reading the gathered records in place
            self._globalise_into(ex.out("b_doc"), b_doc, self.doc_base)
            ex.out("b_score").copy_(b_score)
            ex.out("b_n").copy_(b_n)
            # (asynchronous: the collective runs on the backend's own stream beside what is enqueued next -- stage 2's pass over
            # the shard; stage 1 is finished from inside _dense, right behind that pass)
            gathering = dist.all_gather_into_tensor(ex.recv, ex.send, group=self.group, async_op=True)
            if rerank:
                rx = ex.rerank.get((k1, keep))
                if rx is None:
                    rx = ex.rerank[(k1, keep)] = _RerankExchange(self.world, Q, k1, keep, b_doc.device)
            bounds = self._shard_bounds(b_doc.device) if rerank and self.a2a == "records" and hasattr(e, "rerank_plan") else None
            stage1 = {}

            def finish_stage1():
                gathering.wait()
                stage1["lists"] = e.merge_gathered(ex, "b_doc", "b_score", "b_n", None, k1)[:3]
                if bounds is not None:
                    # who owns which candidate: counted now, copied to the host while the rest of stage 2 runs
                    stage1["plan"] = rx.records()
                    e.rerank_plan(stage1["lists"][0], stage1["lists"][2], bounds, self.rank, rx.Qs, stage1["plan"])
                    stage1["plan"].to_host()
        else:
            b_doc = self._globalise(b_doc, self.doc_base)
            finish_stage1 = None
        d_doc, d_score, d_chunk, d_n = self._dense(qvec, k2, max_chunks_per_doc, dense_batched, beside=finish_stage1)
        if self.world > 1:
            b_doc, b_score, b_n = stage1["lists"]
            plan = stage1.get("plan")
            # stage 2: the same; the arg-max chunk of a dense entry travels with it as the merge payload
            self._globalise_into(ex_d.out("d_doc"), d_doc, self.doc_base)
            ex_d.out("d_score").copy_(d_score)
            self._globalise_into(ex_d.out("d_chunk"), d_chunk, self.row_base)
            ex_d.out("d_n").copy_(d_n)
            dist.all_gather_into_tensor(ex_d.recv, ex_d.send, group=self.group)
            d_doc, d_score, d_n, d_chunk = e.merge_gathered(ex_d, "d_doc", "d_score", "d_n", "d_chunk", k2)
        else:
            d_doc = self._globalise(d_doc, self.doc_base)
            d_chunk = self._globalise(d_chunk, self.row_base)
        out = dict(bm25=(b_doc, b_score, b_n), dense=(d_doc, d_score, d_chunk, d_n))
        if not rerank:
            return out
        if self.world == 1:
            cos, meta = e.rerank_gather(qvec, b_doc, b_n, doc_base=self.doc_base, row_base=self.row_base, max_chunks=max_chunks)
            r = e.rerank_fuse(b_doc, b_score, b_n, cos, meta, **rerank_params)
            out["rerank"] = r if keep == k1 else self._truncate(r, keep)
            return out
        Qs = rx.Qs
        lo, hi = min(Q, self.rank * Qs), min(Q, (self.rank + 1) * Qs)
        if plan is not None:
            # a. my documents' records of every query, ordered by the rank that owns the query: ONE gather launch
            e.rerank_gather_records(qvec, b_doc, b_n, plan, rx.rec_send, doc_base=self.doc_base, row_base=self.row_base,
                                    max_chunks=max_chunks)
            # b. every record to the owner of its query (the host reads the split sizes here: copied before stage 2 ran)
            send_splits, recv_splits = plan.splits(self.rank)
            dist.all_to_all_single(rx.rec_recv[:sum(recv_splits)], rx.rec_send[:sum(send_splits)], recv_splits, send_splits,
                                   group=self.group)
        else:
            # a. / b. the dense form: my documents' halves of every query, written into the block of the rank that owns the
            #    query (the rows of a short last block stay zero), every half to the owner of its query
            e.rerank_gather_blocks(qvec, b_doc, b_n, rx.a2a_send.view(self.world, rx.block), Qs, doc_base=self.doc_base,
                                   row_base=self.row_base, max_chunks=max_chunks)
            dist.all_to_all_single(rx.a2a_recv, rx.a2a_send, group=self.group)
        # c. join + the float64 chain, for my queries only
        if hi > lo:
            if plan is not None:
                cos, meta = e.rerank_scatter(rx.rec_recv, plan, lo, hi - lo, k1)
            else:
                cp, mp = rx.recv_parts()
                cos, meta = e.rerank_combine(cp, mp, hi - lo)
            fused = e.rerank_fuse(b_doc[lo:hi], b_score[lo:hi], b_n[lo:hi], cos, meta, **rerank_params)
            if keep < k1:
                fused = self._truncate(fused, keep)
            for (name, _, per_k), x in zip(_FUSED, fused):
                rx.out_view(name)[:hi - lo].copy_(x)
        # d. the fused lists of all queries to every rank
        dist.all_gather_into_tensor(rx.out_recv, rx.out_send, group=self.group)
        out["rerank"] = tuple(rx.gathered(name)[:Q] for name, _, _ in _FUSED)
        return out
