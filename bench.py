#!/usr/bin/env python3
"""Headline benchmark: queries/sec (and single-query p50 latency) of the two-stage retriever, top-100 on a
synthetic 1 M-document / 5 M x 768 f32-chunk corpus resident in HBM (BASELINE.json metric / configs[2]).

A STEP = one pass of the whole hot path over one batch of `--queries-per-step` (default 256) queries:
    stage 1  BM25 term-at-a-time + top-1000                        (msr_bm25_topk)
    stage 2  dense full scan: q x chunk cosine, per-doc max-pool, top-100   (msr_dense_topk; ONE pass over E per 256 queries)
    fuse     reference rerank chain on the stage-1 candidates -> top-100     (msr_rerank_gather + _fuse)
With N > 1 GPUs the corpus is doc-sharded and, by default, the batch is 256 queries PER GPU: every GPU sweeps 1/N of
the rows for N times the queries, i.e. the same work per GPU and step at every N -- reported as "scaling": "weak"
(an explicit --queries-per-step fixes the batch instead: "strong").  Per step one all-gather of the per-shard top-k
lists, one all-to-all of the candidates' cosines (every query's halves to the rank that fuses it) and one all-gather of the
fused lists cross xGMI.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dense-scan kernel: algorithmic bytes per launch / mean launch duration (hipEvents recorded on
                the launch stream inside the timed region) against the 8 TB/s HBM peak
  cpu_baseline  the C restatement in oracle/ (kind "port") timed on the host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_shard(args, rank, world, dev, n_queries):
    from msretr.synthetic import SEED, synthetic_corpus, synthetic_queries, synthetic_query_terms, synthetic_shard
    t0 = time.time()
    if world > 1 and not args.verify:
        # only THIS rank's document range of the postings is ever materialised (the token stream is generated block by
        # block; document frequencies -- idf, avgdl, the query pool -- are counted over all blocks, so they are the
        # global ones on every rank); embeddings: this rank's rows only (i.i.d. rows, a shard-local stream)
        shard, df = synthetic_shard(args.docs, args.chunks, args.terms, rank, world, seed=SEED, device=dev)
        terms = synthetic_query_terms(df, n_queries, seed=777)
        full = None
    else:
        full = synthetic_corpus(args.docs, n_chunks=args.chunks, n_terms=args.terms, seed=SEED, device=dev,
                                with_embeddings=args.verify)
        # query terms come from the GLOBAL document frequencies, so every rank draws the same queries
        terms, _ = synthetic_queries(full if not args.verify else _without_emb(full), n_queries, seed=777, device="cpu")
        shard = full.shard(rank, world) if world > 1 else full
        if args.verify:                               # rehearsal mode: shards are slices of ONE global matrix
            return shard, terms, full
    C = shard.n_chunks
    if C == 0:
        return shard, terms, None
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + 7919 * (rank + 1))
    emb = torch.empty((C, 768), dtype=torch.float32, device=dev)
    blk = 1 << 18
    for s in range(0, C, blk):
        x = torch.randn((min(blk, C - s), 768), generator=g, device=dev)
        emb[s:s + len(x)] = x / x.norm(dim=1, keepdim=True)
    shard.emb = emb
    torch.cuda.synchronize()
    log(f"[rank {rank}] corpus: {shard.n_docs} docs, {C} chunks, {int(shard.post_doc.numel())} postings "
        f"(doc_base {shard.doc_base}) in {time.time() - t0:.1f}s")
    return shard, terms, None


def _without_emb(ix):
    import copy
    c = copy.copy(ix)
    c.emb = None
    return c


def verify_against_unsharded(args, full, terms, qvec, out_sharded, dev):
    """Rehearsal check (small corpora): the sharded step must reproduce an unsharded engine bit for bit."""
    from msretr.distributed import ShardedEngine
    from msretr.engine import DeviceEngine
    Q = args.queries_per_step
    eng = DeviceEngine(full, device=dev.index, max_queries=Q, max_k=max(args.k1, args.k2), rerank_max_docs=args.k1)
    ref = ShardedEngine(eng, 0, 0)
    ref.world = 1                                     # no collectives: this is the single-engine reference
    exp = ref.search([full.term_ids(t) for t in terms[:Q]], qvec[:Q], k1=args.k1, k2=args.k2)
    for key in exp:
        for a, b in zip(out_sharded[key], exp[key]):
            if a is None or b is None:
                continue
            assert torch.equal(a, b), f"sharded != unsharded for {key}"
    eng.close()
    return True


def make_query_vectors(n, dev, seed):
    """Same vectors on every rank: seeded CPU stream, norm in [5, 15] (the encoder output is not normalised)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    v = torch.randn((n, 768), generator=g)
    v = v / v.norm(dim=1, keepdim=True) * torch.empty(n, 1).uniform_(5.0, 15.0, generator=g)
    return v.to(dev)


def measured_traffic(args, world, kernel):
    """HBM bytes per launch of the roofline kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE are collected in separate runs of this same command and corrected as
    MI355X_MICROARCH.md prescribes; see tools/summarize_prof.py).  None if no matching measurement exists."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    for e in json.load(open(path)):
        if (e["kernel"] == kernel and e["n_chunks"] == args.chunks and e["n_docs"] == args.docs and
                e["queries_per_step"] == args.queries_per_step and e["workload"] == args.workload):
            return e["hbm_bytes_per_launch"]
    return None


def _p50(xs):
    return 1e3 * float(np.median(xs))


def cpu_baseline(args, shard, terms, qvec, dev):
    """The CPU restatements of the path (oracle/) timed on this box's host cores, as BASELINE.md section 2 lays it out:
      port        C + OpenMP: BM25 over ALL postings, dense cosine / max-pool / top-k over ALL chunk rows, + the rerank / fuse
                  chain (numpy) on the stage-1 candidates -- the fastest honest CPU form of the whole step; `value`
      vectorised  NumPy: CSR BM25, E @ q over all rows, np.maximum.reduceat per document, argpartition top-k
      literal     the reference's own shape: pure-Python dict / loop BM25 on a 100 k-document corpus and the pandas
                  iterrows / groupby rerank chain on 1000 candidates x <= 10 chunks, 1 thread
    >= 20 timed queries per leg after 3 warm-ups, p50 per query.  Returns (cpu_baseline object, port BM25 results, port
    dense results, port rerank results): the results of the timed queries, against which the GPU path's parity is reported."""
    from oracle import bm25_ref, c_oracle, dense_ref, rerank_ref
    from msretr.synthetic import synthetic_corpus, synthetic_queries
    threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    c_oracle.set_threads(threads)
    nq, warm = max(1, args.cpu_queries), 3
    t0 = time.time()
    doc_off = shard.doc_off.cpu().numpy().astype(np.int32)
    emb = np.empty((shard.n_chunks, 768), np.float32)
    blk = 1 << 19
    for s0 in range(0, shard.n_chunks, blk):               # block-wise: no second full-size temporary on the host
        emb[s0:s0 + blk] = shard.emb[s0:s0 + blk].cpu().numpy()
    ix = {k: np.ascontiguousarray(getattr(shard, k).cpu().numpy()) for k in ("term_off", "post_doc", "post_tf", "doc_len", "idf")}
    ix["avgdl"] = shard.avgdl
    log(f"[cpu baseline] host copy of {int(ix['post_doc'].size)} postings + {shard.n_chunks} chunk rows: {time.time() - t0:.1f}s")
    q_host = qvec[:nq + warm].cpu().numpy()
    chunk_doc = np.repeat(np.arange(shard.n_docs, dtype=np.int64), np.diff(doc_off.astype(np.int64)))
    tb, td, tr, results, dense_results, rerank_results = [], [], [], [], [], []
    for i in range(nq + warm):                              # ---- port
        ut, qtf = bm25_ref.prepare_query(terms[i], ix["term_off"])
        t1 = time.perf_counter()
        r = c_oracle.bm25_topk(ix, ut, qtf, args.k1, 0.0, shard.k1, shard.b)
        t2 = time.perf_counter()
        dres = c_oracle.dense_topk(emb, doc_off, q_host[i], args.k2)
        t3 = time.perf_counter()
        # rerank / fuse of the stage-1 candidates: first <= 10 chunks each, cosine, min-max, blend, positional, arg-max, sort
        # (oracle/rerank_ref.py: the restatement pinned to the reference's per-stage outputs, tests/golden/rerank_chain.json)
        cd_ = [(int(d), float(b), int(doc_off[d]), min(int(doc_off[d + 1] - doc_off[d]), 10)) for d, b in zip(r[0], r[1])]
        cd_ = [c for c in cd_ if c[3] > 0]
        best = []
        if cd_:
            ridx = np.concatenate([np.arange(lo, lo + n, dtype=np.int64) for _, _, lo, n in cd_])
            cos = rerank_ref.cosine_f32(q_host[i], emb[ridx])
            pooled, _ = rerank_ref.chain_from_cosines([c[0] for c in cd_], [c[3] for c in cd_], [c[1] for c in cd_], cos)
            lo_of = {c[0]: c[2] for c in cd_}
            best = sorted(((d, s_, o_, lo_of[d] + b_) for d, s_, o_, b_ in pooled), key=lambda x: (-x[1], x[0]))
        t4 = time.perf_counter()
        if i >= warm:
            tb.append(t2 - t1); td.append(t3 - t2); tr.append(t4 - t3); results.append(r); dense_results.append(dres)
            rerank_results.append(best)
    port = {"queries": nq, "cores": threads, "p50_ms": _p50([a + b + c for a, b, c in zip(tb, td, tr)]),
            "bm25_p50_ms": _p50(tb), "dense_p50_ms": _p50(td), "rerank_p50_ms": _p50(tr),
            "what": f"C + OpenMP ({threads} threads): BM25 top-{args.k1} over all {int(ix['post_doc'].size)} postings, cosine / per-document "
                    f"max / top-{args.k2} over all {shard.n_chunks} chunk rows; rerank / fuse chain of the stage-1 candidates in numpy"}
    port["value"] = 1e3 / port["p50_ms"]
    tv = []
    for i in range(nq + warm):                              # ---- vectorised NumPy
        t1 = time.perf_counter()
        bm25_ref.topk(ix, terms[i], args.k1, 0.0, shard.k1, shard.b)
        qn = q_host[i] / max(np.linalg.norm(q_host[i]), 1e-30)
        cos = emb @ qn                                       # rows are unit-norm: this IS the cosine (BLAS, all cores)
        best = np.maximum.reduceat(cos, doc_off[:-1].astype(np.int64))
        best[np.diff(doc_off) == 0] = -np.inf
        top = np.argpartition(-best, args.k2)[:args.k2]
        top = top[np.argsort(-best[top], kind="stable")]
        if i >= warm:
            tv.append(time.perf_counter() - t1)
    vec = {"queries": nq, "cores": os.cpu_count(), "p50_ms": _p50(tv), "value": 1e3 / _p50(tv),
           "what": "NumPy: CSR BM25 (float64, reference operation order), E @ q over all rows (BLAS threads = all cores), "
                   "np.maximum.reduceat per document, argpartition top-k"}
    lit = None
    try:                                                    # ---- literal (reference-shaped), its own 100 k-document corpus
        small = synthetic_corpus(100_000, n_chunks=0, n_terms=200_000, seed=11, device=dev)
        st, _ = synthetic_queries(small, nq + warm, seed=12, device="cpu")
        sx = {k: np.ascontiguousarray(getattr(small, k).cpu().numpy()) for k in ("term_off", "post_doc", "post_tf", "doc_len", "idf")}
        sx["avgdl"] = small.avgdl
        rng = np.random.default_rng(3)
        tl, tlr = [], []
        for i in range(nq + warm):
            t1 = time.perf_counter()
            ids, sc = bm25_ref.topk_literal(sx, st[i], args.k1)
            t2 = time.perf_counter()
            n_ch = rng.integers(1, 11, size=len(ids))
            e = rng.standard_normal((int(n_ch.sum()), 768)).astype(np.float32)
            e /= np.linalg.norm(e, axis=1, keepdims=True)
            rows, p = [], 0
            for d, n in zip(ids.tolist(), n_ch.tolist()):
                rows += [(d, p + j, e[p + j]) for j in range(n)]
                p += n
            t3 = time.perf_counter()
            if rows:
                rerank_ref.rerank_chain_pandas(rows, q_host[i], ids.tolist(), sc.tolist())
            t4 = time.perf_counter()
            if i >= warm:
                tl.append(t2 - t1); tlr.append(t4 - t3)
        lit = {"queries": nq, "cores": 1, "p50_ms": _p50([a + b for a, b in zip(tl, tlr)]), "bm25_p50_ms": _p50(tl),
               "rerank_p50_ms": _p50(tlr),
               "what": "reference-shaped Python: dict / loop BM25 on 100000 documents (bm25_indexer.py:450-485 shape) + pandas "
                       "merge / iterrows / groupby rerank chain on <= 1000 candidates x <= 10 chunks (reranker_api.py:357-372 shape)"}
        lit["value"] = 1e3 / lit["p50_ms"]
    except Exception as ex:
        lit = {"error": repr(ex)}
    obj = {"value": port["value"], "unit": "queries/sec", "cores": threads, "kind": "port", "queries": nq,
           "p50_ms": port["p50_ms"],
           "sample": f"{nq} queries (after {warm} warm-ups) of the benchmark's own query pool, whole corpus, per-query p50; see "
                     "port / vectorised / literal",
           "port": port, "vectorised": vec, "literal": lit}
    return obj, results, dense_results, rerank_results


def dense_parity(gpu, cpu, k):
    """max |score difference| <= 1e-5 and top-k document equality of the GPU dense stage against the CPU
    restatement, query by query.  gpu: (doc, score, n) arrays of the GPU call [Q, k]; cpu: list of (doc, score, chunk) of the
    oracle for the same queries.  Two lists may differ in documents only where scores are within rounding of each other (a
    near-tie swaps two ranks, or, at rank k, swaps a document in or out): a document found on one side only must score within
    2e-5 of that side's k-th score."""
    gd, gs, gn = gpu
    worst, same_rank, sets_ok, n_ok = 0.0, 0, True, True
    for i, (cd, cs, _) in enumerate(cpu):
        n = int(gn[i])
        n_ok = n_ok and n == len(cd)
        m = min(n, len(cd))
        if m == 0:
            continue
        worst = max(worst, float(np.abs(gs[i, :m] - cs[:m]).max()))          # rank by rank: the sorted score lists agree
        same_rank += int((gd[i, :m] == cd[:m]).sum())
        g_only = set(gd[i, :n].tolist()) - set(cd.tolist())
        c_only = set(cd.tolist()) - set(gd[i, :n].tolist())
        if g_only or c_only:
            gsc = dict(zip(gd[i, :n].tolist(), gs[i, :n].tolist()))
            csc = dict(zip(cd.tolist(), cs.tolist()))
            kth = float(cs[m - 1])
            sets_ok = sets_ok and all(abs(gsc[d] - kth) <= 2e-5 for d in g_only) and all(abs(csc[d] - kth) <= 2e-5 for d in c_only)
    total = sum(len(c[0]) for c in cpu)
    return {"queries": len(cpu), "k": k, "max_abs_score_diff": worst, "within_1e-5": worst <= 1e-5 and n_ok,
            "top_k_doc_sets_equal_up_to_boundary_near_ties": bool(sets_ok),
            "same_doc_at_same_rank": same_rank / max(1, total),
            "kernel": "the batch of the timed steps in one msr_dense_topk call (streaming pass + exact f32 rescoring), rows of the CPU sample"}


def _term_name(t):
    """an alphabetic name for synthetic term id t (so that the facade's default tokeniser -- lower-cased alphabetic tokens,
    the stand-in for the reference's spaCy lemmatiser -- reads it back): 'q' + base-26 digits."""
    s = ""
    t = int(t)
    while True:
        s = chr(97 + t % 26) + s
        t //= 26
        if t == 0:
            return "q" + s


def facade_bench(args, shard, eng, terms, qvec, n_queries=1024, single=20):
    """The drop-in API a maintainer of the reference calls, timed end to end on the HOST clock: `Retriever.batch_search_to_file`
    (search_api.py:331-367: queries.txt -> preprocess_query -> tokenise -> BM25 top-1000 -> rerank -> diversification ->
    `qnum<TAB>rank<TAB>url<TAB>score` lines in a file) for n_queries queries.txt-shaped queries, and `Retriever.search` (the
    /api/search path, one query, UI dicts) for the p50.  The synthetic corpus gets URL / title / text strings (5003 domains)
    for this; query strings are the benchmark's term lists spelled as words (the city term is appended by preprocess_query,
    as in the reference); query vectors are given (the headline's queries arrive as vectors too)."""
    import tempfile
    from msretr.retriever import Retriever
    from msretr.text import CITY
    t0 = time.time()
    N = shard.n_docs
    ids = shard.doc_ids.cpu().numpy() if hasattr(shard.doc_ids, "cpu") else np.asarray(shard.doc_ids)
    pool = ["Synthetic page text. " * (2 + j % 17) for j in range(64)]
    shard.urls = [f"https://site{int(d) % 5003}.example/page/{int(d)}" for d in ids]
    shard.titles = [f"Title {int(d)}" for d in ids]
    shard.texts = [pool[i & 63] for i in range(N)]
    shard._url_group = None
    used = sorted({int(t) for tl in terms[:n_queries] for t in tl})
    shard.vocab = {_term_name(t): t for t in used if t != 0}
    shard.vocab[CITY] = 0
    rt = Retriever(indexer=eng, freeze_gc=True)      # (the corpus tables are permanent: no full collection walks them mid-batch)
    # (the engine was bound before the URL strings existed: every document is its own URL group, which is what these URLs
    # say too -- one page per document)
    texts = [" ".join(_term_name(t) for t in tl if t != 0) for tl in terms[:n_queries]]
    embs = qvec[:n_queries].cpu().numpy()
    tmp = tempfile.mkdtemp(prefix="msr_facade_")
    qf, of = os.path.join(tmp, "queries.txt"), os.path.join(tmp, "batch_search_results.txt")
    with open(qf, "w", encoding="utf-8") as f:
        for i, t in enumerate(texts):
            f.write(f"{i + 1}\t{t}\n")
    setup_s = time.time() - t0
    from msretr.text import read_queries_file
    nq = read_queries_file(qf)
    rt.batch_search(nq[:64], query_embeddings=embs[:64]).text()          # warm-up (binds the domain table, builds the URL blob)
    torch.cuda.synchronize()
    # timed runs, each with the wall time the host spends in every stage of the pipelined call; the fastest run is reported
    acc = {}

    def timed(obj, name, label):
        fn = getattr(obj, name)

        def wrap(*a_, **k_):
            t_ = time.perf_counter()
            try:
                return fn(*a_, **k_)
            finally:
                acc[label] = acc.get(label, 0.0) + time.perf_counter() - t_
        setattr(obj, name, wrap)
        return fn
    saved = [(rt, "_prepare", timed(rt, "_prepare", "preprocess + tokenise + term ids + vectors")),
             (rt, "_enqueue_chunk", timed(rt, "_enqueue_chunk", "pack terms + H2D + enqueue kernels and copies")),
             (rt, "_collect_chunk", timed(rt, "_collect_chunk", "wait for a chunk's final rows (GPU not done yet) + copy out")),
             (rt._formatter, "format", timed(rt._formatter, "format", "native line formatting"))]
    times, runs = [], []
    for rep in range(7):
        acc.clear()
        t1 = time.perf_counter()
        n_lines = rt.batch_search_to_file(qf, of, query_embeddings=embs)
        times.append(time.perf_counter() - t1)
        runs.append({k: 1e3 * v for k, v in acc.items()})
    for obj, name, fn in saved:
        setattr(obj, name, fn)
    total = min(times)
    host_ms = runs[times.index(total)]
    host_ms["whole call"] = 1e3 * total
    host_ms["all runs, whole call"] = [round(1e3 * t, 2) for t in times]
    # a longer file (the same queries four times over, 16 chunks): what the pipeline sustains once it is full
    qf4 = os.path.join(tmp, "queries_x4.txt")
    with open(qf4, "w", encoding="utf-8") as f:
        for i in range(4 * len(texts)):
            f.write(f"{i + 1}\t{texts[i % len(texts)]}\n")
    embs4 = np.concatenate([embs] * 4)
    t4 = []
    for rep in range(3):
        t1 = time.perf_counter()
        n4 = rt.batch_search_to_file(qf4, of + ".x4", query_embeddings=embs4)
        t4.append(time.perf_counter() - t1)
    sustained = {"queries": 4 * len(texts), "value": 4 * len(texts) / min(t4), "unit": "queries/sec", "ms_per_batch": 1e3 * min(t4),
                 "lines_written": int(n4)}
    # the same stages one after the other, not pipelined
    t1 = time.perf_counter()
    idl, qv = rt._prepare([q for _, q in nq], embs, None)
    t2 = time.perf_counter()
    doc, score, _, n = rt.final_lists(idl, qv)
    t3 = time.perf_counter()
    from msretr.retriever import BatchLines
    BatchLines([a for a, _ in nq], doc, score, n, rt.index.urls, rt._formatter).write(of)
    t4 = time.perf_counter()
    # the engine's share alone: the same calls with the inputs packed and resident, device time only
    packed = [eng.pack_queries(idl[a:a + 256]) for a in range(0, n_queries, 256)]
    qd = qvec[:n_queries].contiguous()

    def engine_only():
        for j, a in enumerate(range(0, n_queries, 256)):
            b = eng.bm25_topk(None, k=1000, packed=packed[j])
            cos, meta = eng.rerank_gather(qd[a:a + 256], b[0], b[2])
            eng.diversify(eng.rerank_fuse(b[0], b[1], b[2], cos, meta))
    engine_only(); torch.cuda.synchronize()
    t5 = time.perf_counter()
    for _ in range(3):
        engine_only()
    torch.cuda.synchronize()
    eng_ms = 1e3 * (time.perf_counter() - t5) / 3
    # the list-of-dicts form of the same results (search_api.py:276-292 builds it per result in Python): materialised on demand
    t6 = time.perf_counter()
    as_list = list(rt.batch_search(nq, query_embeddings=embs))
    t7 = time.perf_counter()
    lat = []
    for rep in range(2):
        for i in range(single):
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            docs = rt.search(texts[i], query_embedding=embs[i])
            if rep:
                lat.append(time.perf_counter() - t8)
    with open(of, "rb") as f:
        head = f.readline().decode().rstrip("\n")
    ok = n_lines == int(n.sum()) and len(as_list) == n_lines and as_list[0]["formatted_line"] == head and len(docs) > 0
    return {"api": "Retriever.batch_search_to_file (search_api.py:331-367 /api/batch_search_file): queries.txt -> preprocess_query -> "
                   "tokenise -> BM25 top-1000 -> rerank/fuse -> diversification -> top-100 lines written to a file; host clock, "
                   "everything included",
            "queries": n_queries, "value": n_queries / total, "unit": "queries/sec", "ms_per_batch": 1e3 * total,
            "lines_written": int(n_lines), "results_per_query": float(n.mean()),
            "pipelined": "chunks of 256 queries, three stages side by side: the calling thread prepares and enqueues chunk i + 1, the GPU "
                         "ranks chunk i, a second host thread collects, formats (native code) and writes chunk i - 1; the stage times "
                         "below are wall time per stage summed over the chunks, on whichever thread ran them",
            "host_ms_inside_the_pipelined_call": host_ms,
            "longer_file": sustained,
            "stage_ms_unpipelined": {"read + preprocess + tokenise + term ids + vectors": 1e3 * (t2 - t1),
                             "device path incl. packing, H2D, D2H of the final rows": 1e3 * (t3 - t2),
                             "native line formatting + file write": 1e3 * (t4 - t3)},
            "engine_ms_same_batch": eng_ms, "ratio_to_engine": 1e3 * total / eng_ms,
            "engine_calls": "msr_bm25_topk + msr_rerank_gather + msr_rerank_fuse + msr_diversify, 256 queries per call, inputs resident",
            "as_list_of_dicts_ms": 1e3 * (t7 - t6),
            "p50_latency_ms_single_query": 1e3 * float(np.median(lat)),
            "single_query_api": "Retriever.search (search_api.py:69-152 /api/search without the LLM call): UI dicts of the top 100",
            "outputs_sane": bool(ok), "setup_s": setup_s,
            "note": "the reference's live path has no dense full scan (BM25 -> rerank of the 1000 candidates); the headline step adds "
                    "one, so its engine time is not this path's"}


def rerank_parity(gpu, cpu, tol=5e-6):
    """The fused lists of the GPU step (msr_rerank_gather + msr_rerank_fuse on the GPU's own stage-1 candidates) against the
    CPU restatement of reranker_api.py:357-372 on ITS stage-1 candidates (bit-equal lists: bm25_parity_vs_cpu), query by query:
    same documents, new_similarity within `tol` rank by rank (the f64 chain sees f32 cosines that differ in the last bit
    between sklearn's and the kernel's summation order, amplified by the min-max division), normalised BM25 of the winning row
    within 1e-12, and the same document at the same rank except where neighbouring scores are within 2 tol.
    gpu: (doc, score, orig, chunk, n, rows) numpy arrays; cpu: per query the list of (doc, score, orig, row) in rank order."""
    gd, gs, go, gc, gn, _ = gpu
    worst, worst_o, same_rank, total, sets_ok, order_ok, chunk_same = 0.0, 0.0, 0, 0, True, True, 0
    for i, exp in enumerate(cpu):
        n = int(gn[i])
        sets_ok = sets_ok and n == len(exp) and sorted(gd[i, :n].tolist()) == sorted(e[0] for e in exp)
        m = min(n, len(exp))
        if m == 0:
            continue
        es = np.array([e[1] for e in exp[:m]])
        worst = max(worst, float(np.abs(gs[i, :m] - es).max()))
        emap = {e[0]: e for e in exp}
        for j in range(m):
            d = int(gd[i, j])
            e = emap.get(d)
            if e is None:
                continue
            worst_o = max(worst_o, abs(float(go[i, j]) - e[2]))
            chunk_same += int(int(gc[i, j]) == e[3])
            if d == exp[j][0]:
                same_rank += 1
            elif abs(exp[j][1] - e[1]) > 2 * tol:
                order_ok = False
        total += m
    return {"queries": len(cpu), "entries": total, "max_abs_new_similarity_diff": worst, "within_tol": worst <= tol, "tol": tol,
            "max_abs_original_similarity_diff": worst_o, "doc_sets_equal": bool(sets_ok),
            "order_equal_up_to_near_ties": bool(order_ok), "same_doc_at_same_rank": same_rank / max(1, total),
            "same_winning_chunk": chunk_same / max(1, total),
            "reference": "reranker_api.py:285,357-372 restated in oracle/rerank_ref.py (sklearn-form f32 cosine, Python-float chain)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--chunks", type=int, default=5_000_000)
    ap.add_argument("--terms", type=int, default=1_000_000)
    ap.add_argument("--queries-per-step", type=int, default=0,
                    help="queries per step (the dense stage reads E once per 256 of them); 0 = 256 per GPU: with the "
                         "corpus sharded N ways and the batch N times larger, every GPU does the same work per step "
                         "at every N (weak scaling); an explicit value keeps the batch fixed (strong scaling)")
    ap.add_argument("--k1", type=int, default=1000, help="stage-1 candidates (config.py:13)")
    ap.add_argument("--k2", type=int, default=100, help="final top-k (reranker/config.yaml:30)")
    ap.add_argument("--scan-layout", type=int, default=0)
    ap.add_argument("--scan-variant", type=int, default=0)
    ap.add_argument("--cpu-queries", type=int, default=20, help="timed queries per CPU baseline leg (after 3 warm-ups)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="OpenMP threads of the CPU baseline (a 1-GPU box is entitled to 16 host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timed loop with the bf16 candidate sweep")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="ONE GPU does the per-rank GPU work of an N-GPU step (run with 1/N of --docs/--chunks and N times the "
                         "queries): local stages, the merges of N gathered lists, the rerank gather of the 1/N of the candidates "
                         "a rank owns, the join of N halves and the fuse of 1/N of the queries; no collective is executed")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the steps with stage 1 of the next batch overlapped with the tail of this one (variant_pipelined)")
    ap.add_argument("--with-encoder", action="store_true",
                    help="time the variant that starts from token ids (QueryEncoder -> hybrid step) even with --no-variants")
    ap.add_argument("--facade", action="store_true",
                    help="time the drop-in API (Retriever.batch_search_to_file / Retriever.search) even with --no-variants")
    ap.add_argument("--latency-queries", type=int, default=20)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (with --backend gloo), to exercise the N>1 path on one GPU")
    ap.add_argument("--verify", action="store_true",
                    help="rehearsal: shards are slices of one global corpus and rank 0 checks the sharded results "
                         "against an unsharded engine (small --docs/--chunks only)")
    ap.add_argument("--workload", choices=["hybrid", "bm25", "dense"], default="hybrid",
                    help="hybrid = the headline (BASELINE configs[2]); bm25 = stage 1 only (configs[1]: use --docs "
                         "100000 --chunks 0 --terms 200000 --queries-per-step 1024 --k1 100); dense = full scan only")
    ap.add_argument("--dense-mode", choices=["f32", "bf16"], default="f32",
                    help="dense / hybrid workloads: bf16 = batched candidates (<= 128 queries per step: one bf16 sweep; more: "
                         "the tiled matrix-core GEMM, 1024 queries per pass) + exact f32 rescoring "
                         "(BASELINE configs[4] shape: use --workload dense --queries-per-step 1024)")
    args = ap.parse_args()
    if args.workload == "bm25":
        args.chunks = 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    auto_batch = args.queries_per_step <= 0
    if auto_batch:
        args.queries_per_step = 256 * world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from msretr.distributed import ShardedEngine
    from msretr.engine import DeviceEngine

    Q = args.queries_per_step
    n_pool = Q * 8
    shard, terms, full = build_shard(args, rank, world, dev, n_pool)
    qvec = make_query_vectors(n_pool, dev, seed=778)
    eng = DeviceEngine(shard, device=local_rank, max_queries=max(Q, 1), max_k=max(args.k1, args.k2),
                       rerank_max_docs=args.k1, scan_layout=args.scan_layout, scan_variant=args.scan_variant)
    se = ShardedEngine(eng, shard.doc_base, shard.row_base)
    if args.workload != "bm25" and args.dense_mode == "bf16":
        eng.enable_bf16()
    batches = []
    for b in range(n_pool // Q):
        tl = [shard.term_ids(t) for t in terms[b * Q:(b + 1) * Q]]
        batches.append((eng.pack_queries(tl), qvec[b * Q:(b + 1) * Q].contiguous()))
    out = {}

    def step(i, one=None):
        packed, qv = one if one is not None else batches[i % len(batches)]
        if args.workload == "hybrid":
            return se.search(None, qv, k1=args.k1, k2=args.k2, packed=packed, dense_batched=args.dense_mode == "bf16")
        if args.workload == "bm25":       # stage 1 only; shards merge their lists like the hybrid path does
            b = eng.bm25_topk(None, k=args.k1, packed=packed)
            if world > 1:
                parts = se._allgather_bytes([se._globalise(b[0], se.doc_base), b[1], b[2]])
                b = eng.merge_topk(*[torch.stack([p[j] for p in parts]) for j in range(3)], args.k1)
            return {"bm25": b}
        d = eng.dense_topk_batched(qv, k=args.k2) if args.dense_mode == "bf16" else eng.dense_topk(qv, k=args.k2)
        if world > 1:
            parts = se._allgather_bytes([se._globalise(d[0], se.doc_base), d[1], d[3]])
            m = eng.merge_topk(*[torch.stack([p[j] for p in parts]) for j in range(3)], args.k2)
            d = (m[0], m[1], None, m[2])
        return {"dense": d}

    if args.emulate_ranks > 1:
        assert world == 1 and args.workload == "hybrid", "--emulate-ranks is a single-process stand-in for the hybrid step"
        from msretr.distributed import _RerankExchange
        NE = args.emulate_ranks
        rx = _RerankExchange(NE, Q, args.k1, args.k2, dev)
        plan = rx.records()
        # the merged candidate lists of an NE-way run: slot m of every list belongs to shard m % NE (shard 0 = this engine:
        # global index = local index + (m % NE) * documents per shard), so this rank owns 1 / NE of every query's candidates
        slot_shard = (torch.arange(args.k1, device=dev) % NE).to(torch.int32).unsqueeze(0)
        e_bounds = (torch.arange(NE + 1, dtype=torch.int64) * shard.n_docs).to(torch.int32).to(dev)
        e_src = torch.arange(NE, dtype=torch.int32, device=dev).view(NE, 1)
        e_gathered = {}

        def step(i, one=None):                          # noqa: F811  (replaces the step above)
            packed, qv = one if one is not None else batches[i % len(batches)]
            b = eng.bm25_topk(None, k=args.k1, packed=packed)
            nq = int(qv.shape[0])
            # the dense stage as a rank of an NE-way run does it: begin, (all-reduce MIN of one float per query -- here the
            # rank's own value: the emulated shards are statistically alike), end with the bound; lists shorter than k2
            split = eng.dense_split_max(args.k2)
            if split > 0 and nq > 64:
                d = (torch.empty((nq, args.k2), dtype=torch.int32, device=dev), torch.empty((nq, args.k2), dtype=torch.float32, device=dev),
                     torch.empty((nq, args.k2), dtype=torch.int32, device=dev), torch.empty((nq,), dtype=torch.int32, device=dev))
                for a0 in range(0, nq, split):
                    a1 = min(nq, a0 + split)
                    if a1 - a0 <= 64:
                        for dst, src in zip(d, eng.dense_topk(qv[a0:a1], k=args.k2)):
                            dst[a0:a1].copy_(src)
                    else:
                        bound = eng.dense_begin(qv[a0:a1], k=args.k2, k_part=(args.k2 + NE - 1) // NE)
                        eng.dense_end(a1 - a0, k=args.k2, bound=bound, out=tuple(t[a0:a1] for t in d))
            else:
                d = eng.dense_topk(qv, k=args.k2)
            # the two merges of NE gathered lists.  Stand-in operands: NE copies of the local lists of the FIRST step of this
            # shape, made once -- in the real step the gathered buffer is filled by the collective, not by this GPU's CUs;
            # what the real step does on the GPU before its all-gathers, writing its lists into the send buffers, is done here too
            if nq not in e_gathered:
                rep = lambda t: t.unsqueeze(0).expand(NE, *t.shape).contiguous()
                e_gathered[nq] = ((rep(b[0]), rep(b[1]), rep(b[2])), (rep(d[0]), rep(d[1]), rep(d[3])),
                                  [torch.empty_like(t) for t in (b[0], b[1], b[2], d[0], d[1], d[2], d[3])])
            g_b, g_d, send = e_gathered[nq]
            for dst, src in zip(send, b + d):
                dst.copy_(src)
            eng.merge_topk(*g_b, args.k1)
            eng.merge_topk(*g_d, args.k2)
            if nq != Q:                                  # (the single-query latency loop: local stages only)
                return {"bm25": b, "dense": d}
            # the rerank exchange in its compact form (distributed.py): count who owns what, the N x N matrix to the host (in
            # the real run that copy is made before the dense stage and read here; the emulation keeps the wait where it is
            # cheapest to reason about: right here, unhidden), records of the owned slots, scatter, fuse for Qs queries
            cand = torch.where(b[0] >= 0, b[0] + slot_shard * shard.n_docs, b[0])
            eng.rerank_plan(cand, b[2], e_bounds, 0, rx.Qs, plan)
            plan.to_host()
            eng.rerank_gather_records(qv, cand, b[2], plan, rx.rec_send)          # one launch, all destinations
            send_splits, recv_splits = plan.splits(0)
            # where the all-to-all would be: what arrives is as large as the plan says; its content -- source g's records for
            # MY queries -- is stood in for by this rank's own records for them, the slots moved to source g's slots
            c = min(recv_splits) // 16
            got = rx.rec_recv[:NE * c * 16].view(NE, c, 16)
            got.copy_(rx.rec_send[:c * 16].view(1, c, 16).expand(NE, c, 16))
            got[:, :, 0] += e_src
            got[:, :, 2] += e_src * shard.n_docs            # (and its URL group: the owner's documents are other documents)
            cos, meta = eng.rerank_scatter(rx.rec_recv, plan, 0, rx.Qs, args.k1)
            r = eng.rerank_fuse(cand[:rx.Qs], b[1][:rx.Qs], b[2][:rx.Qs], cos, meta)
            return {"bm25": b, "dense": d, "rerank": r}

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gemm = args.workload != "bm25" and args.dense_mode == "bf16" and Q > 128 and eng.batch_gemm_ok()
    if gemm:                                              # batches of more than 128 queries: the tiled GEMM (csrc/msr_gemm.hip)
        scan_ms, scan_n = eng.kernel_time_ms(2)           # emit pass: every row tile
        samp_ms, samp_n = eng.kernel_time_ms(3)           # sample pass: every 16th tile
    else:
        scan_ms, scan_n = eng.kernel_time_ms(0) if args.workload != "bm25" else (0.0, 0)
    bm_ms, bm_n = eng.kernel_time_ms(1) if args.workload != "dense" else (0.0, 0)
    dense_width = eng.dense_path() if args.workload != "bm25" and args.dense_mode == "f32" else 0   # of the TIMED calls' kernel
    eng.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # single-query latency (p50), same path with a batch of one
    lat = []
    one = [(eng.pack_queries([shard.term_ids(terms[i])]), qvec[i:i + 1].contiguous()) for i in range(args.latency_queries)]
    for rep in range(2):
        for packed, qv in one:
            fence()
            t1 = time.perf_counter()
            step(0, one=(packed, qv))
            torch.cuda.synchronize()
            if rep:
                lat.append(time.perf_counter() - t1)
    p50_ms = 1e3 * float(np.median(lat)) if lat else None

    # The same steps with the stages of CONSECUTIVE batches overlapped (one GPU, `--pipelined` only: measured at +1.8 %, 59.0 ->
    # 60.1 k queries/s on one box -- the latency-bound BM25 stage and the gather both want every wave slot of the chip, side by
    # side each takes about twice as long; profiles/r04_stream256_experiments.md 3b): stage 1 of
    # batch i + 1 (BM25 + its select: latency-bound, 0.55 ms) runs on a second stream, with its own engine (its own scratch),
    # beside what follows the dense pass of batch i (finish chain, rerank gather, fuse: 1.1 ms) -- it is released by an event
    # behind that pass (msr_dense_topk_begin), so the HBM-bound pass itself keeps the whole chip.  Same kernels, same
    # results; reported NEXT TO the headline.
    pipelined = None
    if args.workload == "hybrid" and args.dense_mode == "f32" and world == 1 and args.emulate_ranks <= 1 and Q > 64 \
            and args.pipelined and eng.dense_split_max(args.k2) >= Q:
        try:
            from msretr.engine import DeviceEngine as _DE
            eng_b = _DE(_without_emb(shard), device=local_rank, max_queries=Q, max_k=args.k1, rerank_max_docs=0)
            s_b = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)

            def stage1(i):                                   # -> (lists, event) ; enqueued on the second stream
                with torch.cuda.stream(s_b):
                    b = eng_b.bm25_topk(None, k=args.k1, packed=batches[i % len(batches)][0])
                    ev = torch.cuda.Event()
                    ev.record(s_b)
                for t_ in b:                                 # (allocated on the second stream, read on the main one)
                    t_.record_stream(main)
                return b, ev

            def run(n, first):
                pend = stage1(first)
                for i in range(first, first + n):
                    qv = batches[i % len(batches)][1]
                    b, ev = pend
                    eng.dense_begin(qv, k=args.k2, k_part=args.k2)          # the passes over the rows
                    behind = torch.cuda.Event()
                    behind.record(main)
                    s_b.wait_event(behind)                                   # stage 1 of the next batch: behind the pass
                    pend = stage1(i + 1)
                    d = eng.dense_end(Q, k=args.k2, bound=None)
                    main.wait_event(ev)
                    cos, meta = eng.rerank_gather(qv, b[0], b[2])
                    r = eng.rerank_fuse(b[0], b[1], b[2], cos, meta)
                main.wait_stream(s_b)                                        # (the look-ahead stage 1 of the batch after the last)
                return {"bm25": b, "dense": d, "rerank": r}
            run(2, 0)
            fence()
            tp = time.perf_counter()
            pout = run(args.steps, args.warmup)
            fence()
            p_el = time.perf_counter() - tp
            last = (args.warmup + args.steps - 1) % len(batches)
            ref = se.search(None, batches[last][1], k1=args.k1, k2=args.k2, packed=batches[last][0])
            same = all(bool(torch.equal(x, y)) for key in ("bm25", "dense", "rerank") for x, y in zip(pout[key], ref[key]))
            pipelined = {"what": "stage 1 (BM25 + select) of batch i + 1 on a second stream and engine beside the finish chain, rerank "
                                 "gather and fuse of batch i; released by an event behind the dense pass",
                         "value": Q * args.steps / p_el, "unit": "queries/sec",
                         "ms_per_step": 1e3 * p_el / args.steps, "equals_default_path_bitwise": same,
                         "note": "the timed region holds steps + 1 BM25 stages (the look-ahead of the batch after the last)"}
            eng_b.close()
        except Exception as ex:
            pipelined = {"error": repr(ex)}

    # The same steps with the dense stage on the batched path (bf16 candidates: a sweep per 128 queries, the tiled GEMM for more, + exact f32
    # rescoring; final scores and top-100 are those of the default path up to f32 rounding).  Reported NEXT TO the
    # headline, never as `value`.
    variant = None
    if args.workload == "hybrid" and args.dense_mode == "f32" and not args.no_variants:
        try:
            eng.enable_bf16()
            vstep = lambda i: se.search(None, batches[i % len(batches)][1], k1=args.k1, k2=args.k2,
                                        packed=batches[i % len(batches)][0], dense_batched=True)
            for i in range(2):
                vstep(i)
            fence()
            tv = time.perf_counter()
            for i in range(args.steps):
                vout = vstep(args.warmup + i)
            fence()
            v_el = time.perf_counter() - tv
            if world > 1:
                t = torch.tensor([v_el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                v_el = float(t.item())
            a_doc, a_score = out["dense"][0], out["dense"][1]
            b_doc, b_score = vout["dense"][0], vout["dense"][1]
            # same documents rank by rank, except where neighbouring scores are within rounding of each other (a near-tie
            # may swap two ranks); every score within 2e-6
            tie = torch.zeros_like(a_doc, dtype=torch.bool)
            near = (torch.diff(a_score, dim=1).abs() <= 4e-6)
            tie[:, 1:] |= near
            tie[:, :-1] |= near
            same = bool(((((a_doc == b_doc) | tie).all()) & ((a_score - b_score).abs() <= 2e-6).all()).item())
            vlat = []                                         # single-query latency on the same path
            for rep in range(2):
                for packed1, qv1 in one[:10]:
                    fence()
                    t1 = time.perf_counter()
                    se.search(None, qv1, k1=args.k1, k2=args.k2, packed=packed1, dense_batched=True)
                    torch.cuda.synchronize()
                    if rep:
                        vlat.append(time.perf_counter() - t1)
            variant = {"dense_stage": "bf16 candidate sweep + exact f32 rescore (msr_dense_topk_bf16)",
                       "value": Q * args.steps / v_el, "unit": "queries/sec", "ms_per_step": 1e3 * v_el / args.steps,
                       "p50_latency_ms_single_query": 1e3 * float(np.median(vlat)) if vlat else None,
                       "top100_equals_default_path_within_2e-6": same,
                       "top100_doc_agreement": float((a_doc == b_doc).float().mean().item())}
        except Exception as ex:
            variant = {"error": repr(ex)}

    # The same steps once more with EVERY dense product in exact f32 (scan_variant 2: v_mfma_f32_16x16x4_f32, bit for bit a
    # k-ordered fmaf chain; 64 queries per sweep on the K-split kernel, which is then bound by the f32 matrix rate).
    # Reported NEXT TO the headline as `roofline_exact_f32`, so that the strict-f32 line is a driver-timed number too.
    exact = None
    if args.workload == "hybrid" and args.dense_mode == "f32" and not args.no_variants and args.scan_variant == 0 \
            and eng.scan_arith() == "f16x2" and shard.n_chunks > 0:
        try:
            eng_x = DeviceEngine(shard, device=local_rank, max_queries=max(Q, 1), max_k=max(args.k1, args.k2),
                                 rerank_max_docs=args.k1, scan_layout=args.scan_layout, scan_variant=2)
            se_x = ShardedEngine(eng_x, shard.doc_base, shard.row_base)
            xstep = lambda i: se_x.search(None, batches[i % len(batches)][1], k1=args.k1, k2=args.k2,
                                          packed=batches[i % len(batches)][0])
            for i in range(2):
                xstep(i)
            fence()
            eng_x.set_timing(True)
            tx = time.perf_counter()
            for i in range(args.steps):
                xout = xstep(args.warmup + i)
            fence()
            x_el = time.perf_counter() - tx
            x_ms, x_n = eng_x.kernel_time_ms(0)
            eng_x.set_timing(False)
            if world > 1:
                t = torch.tensor([x_el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                x_el = float(t.item())
            qx = min(Q, eng_x.scan_width())
            per_ms = x_ms / max(1, x_n)
            flops = 2.0 * 768 * shard.n_chunks * qx
            tf = flops / (per_ms * 1e-3) / 1e12
            mfma_bound = qx > 32
            exact = {"dense_stage": "exact f32 products (scan_variant 2)", "value": Q * args.steps / x_el,
                     "unit": "queries/sec", "ms_per_step": 1e3 * x_el / args.steps, "queries_per_sweep": qx,
                     "kernel": "dense_ksplit_kernel<f32>" if mfma_bound else "dense_scan_v2_kernel<f32>",
                     "ms_per_launch": per_ms, "launches": x_n,
                     "bound": "mfma" if mfma_bound else "hbm",
                     "achieved": tf if mfma_bound else shard.n_chunks * 768 * 4 / (per_ms * 1e-3) / 1e9,
                     "peak": F32_MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
                     "unit_roofline": "TFLOP/s" if mfma_bound else "GB/s",
                     "max_abs_score_diff_vs_default": float((xout["dense"][1] - out["dense"][1]).abs().max().item())}
            exact["frac"] = exact["achieved"] / exact["peak"]
            eng_x.close()
        except Exception as ex:
            exact = {"error": repr(ex)}

    # The whole query path of search_api.py:69-152 in one number: token ids -> query encoder (ModernBERT-base forward + mean
    # pooling, reranker_api.py:137-139,355; random weights, 8 tokens per query) -> the hybrid step above with the vectors it
    # produced.  Reported NEXT TO the headline (the headline's queries arrive as vectors, as BASELINE.json's metric has them).
    with_enc = None
    if args.workload == "hybrid" and args.dense_mode == "f32" and (args.with_encoder or not args.no_variants):
        try:
            from msretr.encoder import QueryEncoder, random_weights
            enc = QueryEncoder(random_weights(seed=5), device=local_rank)
            rng = np.random.default_rng(4242)
            toks = [[rng.integers(0, 50000, size=8).tolist() for _ in range(Q)] for _ in range(len(batches))]

            def estep(i):
                qv = enc.encode(toks[i % len(batches)]) * 9.0       # (the served model's vectors are not unit length either)
                return se.search(None, qv, k1=args.k1, k2=args.k2, packed=batches[i % len(batches)][0])
            for i in range(2):
                estep(i)
            fence()
            te = time.perf_counter()
            for i in range(args.steps):
                eout = estep(args.warmup + i)
            fence()
            e_el = time.perf_counter() - te
            tq = time.perf_counter()
            for i in range(args.steps):
                enc.encode(toks[i % len(batches)])
            torch.cuda.synchronize()
            enc_ms = 1e3 * (time.perf_counter() - tq) / args.steps
            if world > 1:
                t = torch.tensor([e_el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e_el = float(t.item())
            with_enc = {"path": "token ids -> QueryEncoder (ModernBERT-base, 22 layers, random weights, 8 tokens per query, all HIP) -> "
                                "BM25 top-%d + dense full scan + rerank/fuse" % args.k1,
                        "value": Q * args.steps / e_el, "unit": "queries/sec", "ms_per_step": 1e3 * e_el / args.steps,
                        "encoder_ms_per_batch": enc_ms, "queries_per_step": Q,
                        "outputs_sane": bool((eout["dense"][3] == args.k2).all().item())}
            del enc
        except Exception as ex:
            with_enc = {"error": repr(ex)}

    # sanity of the last step's outputs (cheap, outside the timed region)
    ok = True
    if "dense" in out:
        d_doc, d_score, d_chunk, d_n = out["dense"]
        lists_ok = bool((d_n == args.k2).all().item()) if args.emulate_ranks <= 1 else bool((d_n > 0).all().item())   # (an emulated
        #                                             rank returns only what it can contribute to the node's top-k: fewer than k2)
        ok = ok and lists_ok and bool((torch.diff(torch.nan_to_num(d_score, neginf=-1e30), dim=1) <= 0).all().item())
    if "bm25" in out:
        ok = ok and bool((out["bm25"][2] > 0).all().item())
    if "rerank" in out:
        if args.emulate_ranks > 1:                    # (the stand-in for the received records is not aligned query by query:
            #                                           a short list at the end of a source's region may get another query's slots)
            ok = ok and float((out["rerank"][4] > 0).float().mean().item()) >= 0.98
        else:
            ok = ok and bool((out["rerank"][4] > 0).all().item())
    if not ok:                                        # say which list failed
        log("outputs_sane false:", {k: [int((v[-1 if k == "dense" else 2 if k == "bm25" else 4] <= 0).sum().item()), len(v[0])]
                                    for k, v in out.items()})

    verified = None
    if args.verify and world > 1:
        first = step(0)                               # batch 0 again, outside the timed region
        if rank == 0:
            verified = verify_against_unsharded(args, full, terms, qvec, first, dev)
        dist.barrier()
    if rank == 0:
        n_ch = shard.n_chunks
        if args.workload == "bm25":
            # 8 B per posting of the query's terms + 4 B doc_len per document and group of 4 queries (one
            # workgroup scores 4 queries per tile) (+ 12 B per candidate emitted, not counted here: the count is data
            # dependent, so the reported fraction is a lower bound).  Timed: the pass over all tiles (the sample pass over
            # every 16th tile that precedes it is inside the step time, not in this kernel figure)
            tq = batches[0][0][1].long()
            tq = tq[(tq >= 0) & (tq < shard.n_terms)]
            toff = shard.term_off.to(tq.device)
            post_bytes = 8 * int((toff[tq + 1] - toff[tq]).sum().item())
            alg_bytes = post_bytes + (Q + 3) // 4 * 4 * shard.n_docs
            k_ms, k_n, kname = bm_ms, bm_n, "bm25_taat_kernel"
        else:
            bf = args.dense_mode == "bf16"
            # queries served by one pass over the matrix: of the kernel the timed calls actually ran (msr_dense_path)
            width = eng.batch_width() if bf else (dense_width or eng.scan_width())
            q_launch = min(Q, width)
            if not bf and width == 256:
                # the 256-query kernel serves up to 4 groups of 256 queries per launch (they share the rows: HBM once, the
                # XCD's L2 for the other groups; msr_gemm_f32_topk) -- mirrors the engine's choice for this batch size
                groups128 = min(8, max(Q, 128) // 128)
                q_launch = 256 * max(1, min((min(Q, 128 * groups128) + 255) // 256, min(4, groups128 // 2)))
            # launches of several 256-query groups read the engine's f16 image of the rows when it holds one (msr_row_image_state)
            image16 = not bf and width == 256 and q_launch > 256 and eng.row_image_state() == "built"
            alg_bytes = n_ch * 768 * (2 if bf or image16 else 4) + (shard.n_docs + 1) * 4 + q_launch * 768 * 4
            wide_kernel = q_launch > 32                         # 33..64 queries per sweep run on the K-split kernel
            k_ms, k_n = scan_ms, scan_n
            kname = ("dense_ksplit_kernel" if wide_kernel else "dense_scan_v2_kernel") + ("<bf16>" if bf else "")
            if not bf and q_launch > 64:                        # more than 64 queries: one streaming pass over the f32 rows
                kname = "gemm_stream256_kernel<emit>" if width == 256 else "gemm_stream_kernel<emit>"
            if gemm:
                kname = "gemm_stream256_kernel<bf16, emit>"
                alg_bytes = n_ch * 768 * 2 + min(Q, 1024) * 768 * 2     # E (bf16) once per 1024-query pass + the queries
        per_launch_ms = k_ms / max(1, k_n)
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, world, kname),
                "traffic_source": "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on an "
                                  "earlier run (not measured by this process); null: no such measurement for this kernel and workload",
                "algorithmic_bytes_per_launch": alg_bytes, "launches": k_n, "ms_per_launch": per_launch_ms}
        if args.workload != "bm25" and not gemm:
            roof["queries_per_launch"] = q_launch
            if kname.startswith("gemm_stream256") and image16:
                roof["rows_read_from"] = ("the engine's f16 image of the rows (launches of several 256-query groups: the values a one-group "
                                          "launch converts from the f32 rows in registers -- same candidates, same results bit for bit; "
                                          "1536 B per row, which is what algorithmic_bytes_per_launch counts here)")
                # such a launch is bound by the matrix pipes (2 * 768 flop per row and query on v_mfma_f32_16x16x32_f16; the dense
                # f16 peak is the bf16 one), not by HBM: 7.7 GB in ~4 / ~8 ms
                flops = 2.0 * 768 * n_ch * q_launch
                tf = flops / (per_launch_ms * 1e-3) / 1e12
                roof.update({"bound": "mfma", "achieved": tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / BF16_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops,
                             "hbm_GBps_algorithmic": achieved})
            elif kname.startswith("gemm_stream256"):
                state = eng.row_copy_state()
                roof["rows_read_from"] = ("the engine's fragment-order copy of the f32 matrix (the same values, laid out so that a load "
                                          "instruction reads whole cache lines; + 3 % padding rows)" if state == "built" else
                                          f"the caller's row-major f32 matrix (fragment-order copy: {state})")
        if args.workload != "bm25" and args.dense_mode == "f32" and eng.scan_arith() == "f32" and q_launch > 32:
            # exact-f32 products at 64 queries per sweep: v_mfma_f32_16x16x4_f32 runs at the f32 vector rate
            # (157.3 TFLOP/s, MI355X_MICROARCH.md), which binds before HBM does (2 * 768 flop per row and query)
            flops = 2.0 * 768 * n_ch * q_launch
            tf = flops / (per_launch_ms * 1e-3) / 1e12
            roof.update({"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops,
                         "hbm_GBps": achieved})
        if gemm:
            # S = E . Q^T on the matrix cores: 2 * 768 flop per (row, query); queries padded to tiles of 256; the sample pass
            # (every 16th row tile, + 1/16 of the flops) is reported beside it and is NOT counted as useful work
            q_pass = (min(Q, 1024) + 255) // 256 * 256
            flops = 2.0 * 768 * n_ch * q_pass
            tf = flops / (per_launch_ms * 1e-3) / 1e12
            roof.update({"bound": "mfma", "achieved": tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / BF16_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops,
                         "queries_per_launch": q_pass, "hbm_GBps_algorithmic": achieved,
                         "sample_pass_ms_per_launch": samp_ms / max(1, samp_n)})
        if args.workload == "hybrid":
            roof["bm25_taat_ms_per_launch"] = bm_ms / max(1, bm_n)
            # the stage-1 kernel against the same roofline (SURVEY 8d bytes: 8 B per posting of the query's terms + 4 B
            # doc_len per document and query; emitted candidates not counted, so this is a lower bound)
            tq = batches[0][0][1].long()
            tq = tq[(tq >= 0) & (tq < shard.n_terms)]
            toff = shard.term_off.to(tq.device)
            df_q = (toff[tq + 1] - toff[tq])
            bm_bytes = 8 * int(df_q.sum().item()) + (Q + 3) // 4 * 4 * shard.n_docs   # doc_len: once per 4 queries
            bm_gbs = bm_bytes / (roof["bm25_taat_ms_per_launch"] * 1e-3) / 1e9 if bm_n else 0.0
            # what the kernel really moves: the long lists with negative idf are looked up, not streamed (msr_bm25.hip); 12 B per
            # streamed posting + 8 B per table lookup (one per streamed posting and looked-up term, an upper bound)
            idf_q = shard.idf.to(tq.device)[tq]
            pruned = (idf_q < 0) & (df_q >= 2048)
            streamed = int(df_q[~pruned].sum().item())
            roof["bm25_taat"] = {"achieved": bm_gbs, "unit": "GB/s", "frac": bm_gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": bm_bytes, "launches": bm_n,
                                 "note": "SURVEY 8d byte model of a term-at-a-time pass over EVERY posting of the query's terms; the kernel "
                                         "streams only the lists that can raise a score and looks the negative-idf lists up per touched "
                                         "document, so this is an effective rate, not bytes moved",
                                 "streamed_postings_per_launch": streamed, "postings_named_per_launch": int(df_q.sum().item()),
                                 "bytes_moved_per_launch_upper_bound": 32 * streamed,      # posting 12 + lookup 8 + candidate 12
                                 "moved_GBps": (32 * streamed) / (roof["bm25_taat_ms_per_launch"] * 1e-3) / 1e9 if bm_n else 0.0}
        dense_dt = {"f32": "f32", "f16x2": "f32 (exact f32 cosines of every returned document; candidates filtered by one pass of f16 products with a measured margin; batches <= 64 queries: f16x2-split products, |err| <= 8e-6)",
                    "none": "-"}[eng.scan_arith()]
        names = {"hybrid": "two-stage retrieval top-100 (BM25 top-1000 + dense full scan + rerank/fuse)" +
                           (" [dense stage: bf16 candidates + f32 rescore]" if args.dense_mode == "bf16" else ""),
                 "bm25": f"BM25 top-{args.k1}", "dense": f"dense full-scan top-{args.k2}" + (" (bf16 candidates + f32 rescore)" if args.dense_mode == "bf16" else "")}
        line = {
            "metric": "queries/sec, " + names[args.workload],
            "value": Q * args.steps / elapsed, "unit": "queries/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak" if auto_batch else "strong", "vs_baseline": None,
            "dtype": {"hybrid": f"{dense_dt} (dense cosine) / f64 (BM25, fuse)", "bm25": "f64",
                      "dense": "bf16 candidates, f32 final scores" if args.dense_mode == "bf16" else dense_dt}[args.workload],
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.docs} docs / {args.chunks} x 768 f32 chunks / "
                                   f"{int(shard.post_doc.numel()) if world == 1 else 'sharded'} postings, "
                                   f"{Q} queries per step, doc-sharded x{world}",
                       "n_docs": args.docs, "n_chunks": args.chunks, "n_terms": args.terms,
                       "queries_per_step": Q, "k_stage1": args.k1, "k_final": args.k2,
                       "scan_layout": args.scan_layout, "scan_variant": args.scan_variant},
            "p50_latency_ms_single_query": p50_ms, "outputs_sane": ok, "roofline": roof,
        }
        if args.emulate_ranks > 1:
            line["emulated_ranks"] = args.emulate_ranks
            line["metric"] += f" [per-rank GPU work of a {args.emulate_ranks}-GPU step on ONE GPU: no collectives executed]"
        if verified is not None:
            line["sharded_equals_unsharded"] = verified
        if pipelined is not None:
            line["variant_pipelined"] = pipelined
        if variant is not None:
            line["variant_bf16_candidates"] = variant
        if exact is not None:
            line["roofline_exact_f32"] = exact
        if with_enc is not None:
            line["variant_with_encoder"] = with_enc
        if world == 1 and args.workload == "hybrid" and (args.facade or not args.no_variants):
            try:
                line["facade"] = facade_bench(args, shard, eng, terms, qvec, n_queries=min(1024, n_pool))
            except Exception as ex:
                line["facade"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline and args.workload == "hybrid":
            try:
                cb, cres, cdense, crerank = cpu_baseline(args, shard, terms, qvec, dev)
                line["cpu_baseline"] = cb
                # BASELINE.md publishes no number for this metric (vs_baseline stays null); the ratio to the CPU port timed in
                # this very run, on this box's host cores, is reported under its own name
                line["vs_cpu_baseline"] = line["value"] / cb["value"] if cb.get("value") else None
                # BM25 parity of the GPU path against the C restatement on the same queries (bitwise)
                tl = [shard.term_ids(t) for t in terms[3:3 + len(cres)]]   # (the timed queries follow 3 warm-ups)
                gd, gs, gn = [x.cpu().numpy() for x in eng.bm25_topk(tl, k=args.k1)]
                line["bm25_parity_vs_cpu"] = all(
                    gd[i, :gn[i]].tolist() == cres[i][0].tolist() and gs[i, :gn[i]].tolist() == cres[i][1].tolist()
                    for i in range(len(cres)))
                # dense parity: the GPU answers of the SAME call shape the timed steps use -- batch 0, Q queries in
                # one msr_dense_topk call -- against the C restatement for the CPU sample's queries (rows 3 .. 3 + n of it)
                dd, ds, _, dn = [x.cpu().numpy() for x in eng.dense_topk(qvec[:max(Q, 3 + len(cdense))].contiguous(), k=args.k2)]
                sl = slice(3, 3 + len(cdense))
                line["dense_parity_vs_cpu"] = dense_parity((dd[sl], ds[sl], dn[sl]), cdense, args.k2)
                # rerank / fuse parity: the step's own call shape (batch 0 of the timed steps), rows of the CPU sample
                rr = se.search(None, batches[0][1], k1=args.k1, k2=args.k2, packed=batches[0][0])["rerank"] if Q >= 3 + len(crerank) \
                    else se.search([shard.term_ids(t) for t in terms[:3 + len(crerank)]], qvec[:3 + len(crerank)].contiguous(),
                                   k1=args.k1, k2=args.k2)["rerank"]
                line["rerank_parity_vs_cpu"] = rerank_parity([x[sl].cpu().numpy() for x in rr], crerank)
            except Exception as ex:  # the baseline must never take the GPU numbers down with it
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
