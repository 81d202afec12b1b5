"""Seeded synthetic corpus of the shape BASELINE.json names.  There is no network and the
reference ships no database, so benchmarks and most parity tests run on this.  torch is used only as the
array engine (CPU for tests, the GPU for the 1 M-document corpus, where numpy would take minutes).

  documents   doc_id strictly increasing with random gaps; length ~ lognormal(5.6, 0.8) clipped [8, 20000]
  vocabulary  Zipf(s = 1.07) over term ids 1..V-1; term 0 is the stand-in for "tuebingen": forced into
              85 % of the documents so that its idf is negative (log10, no clamp: bm25_indexer.py:138)
  idf         float32(log10((N - df + 0.5) / (df + 0.5))), avgdl float32(mean length)   (REAL columns)
  chunks      1 + Poisson(4) per document clipped to [1, 64], adjusted to hit the requested total
  embeddings  standard normal float32, rows L2-normalised (indexer/indexer.py:165)
  queries     term 0 + 1..4 terms sampled proportional to sqrt(df) from ranks 50..50000, 5 % with a repeated term;
              vector = normalised(random chunk + 0.5 noise) * U[5, 15]  (the encoder output is not
              normalised, reranker_api.py:355)
"""
import math

import numpy as np
import torch

from .index import DIM, CorpusIndex

SEED = 20250815


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def synthetic_corpus(n_docs, n_chunks=None, n_terms=None, seed=SEED, device="cpu", with_postings=True,
                     with_embeddings=True, city_frac=0.85, mean_log_len=5.6, sigma_len=0.8, max_len=20000,
                     emb_block=1 << 18):
    dev = torch.device(device)
    g = _gen(dev, seed)
    N = int(n_docs)
    V = int(n_terms or (200_000 if N <= 200_000 else 1_000_000))
    V = max(V, 64)
    gaps = torch.randint(1, 5, (N,), generator=g, device=dev, dtype=torch.int64)
    doc_ids = torch.cumsum(gaps, 0) + 1000
    ix = CorpusIndex(doc_ids=doc_ids, total_docs=N, n_docs_global=N)
    if with_postings:
        lens = torch.exp(mean_log_len + sigma_len * torch.randn(N, generator=g, device=dev))
        lens = lens.clamp_(8, max_len).to(torch.int64)
        total = int(lens.sum().item())
        w = 1.0 / torch.arange(1, V, device=dev, dtype=torch.float64) ** 1.07
        cdf = torch.cumsum(w, 0)
        cdf = (cdf / cdf[-1]).to(torch.float32)
        start = torch.cumsum(lens, 0) - lens
        key = torch.empty(total, dtype=torch.int64, device=dev)
        blk = 1 << 26
        doc_of = torch.repeat_interleave(torch.arange(N, device=dev), lens)
        for s in range(0, total, blk):
            u = torch.rand(min(blk, total - s), generator=g, device=dev)
            term = torch.searchsorted(cdf, u).clamp_(max=V - 2) + 1
            key[s:s + len(u)] = doc_of[s:s + len(u)] * V + term
        forced = torch.rand(N, generator=g, device=dev) < city_frac
        key[start[forced]] = torch.arange(N, device=dev)[forced] * V          # first token -> term 0
        del doc_of
        key, _ = torch.sort(key)
        uk, tf = torch.unique_consecutive(key, return_counts=True)
        del key
        p_doc, p_term = uk // V, uk % V
        order = torch.argsort(p_term, stable=True)                            # (term, doc) order
        p_doc, p_term, tf = p_doc[order], p_term[order], tf[order]
        df = torch.bincount(p_term, minlength=V)
        term_off = torch.zeros(V + 1, dtype=torch.int64, device=dev)
        term_off[1:] = torch.cumsum(df, 0)
        dff = df.to(torch.float64)
        idf = torch.log10((N - dff + 0.5) / (dff + 0.5)).to(torch.float32)
        idf[df == 0] = 0.0
        ix.doc_len = lens.to(torch.int32)
        ix.term_off, ix.post_doc, ix.post_tf = term_off, p_doc.to(torch.int32), tf.to(torch.int32)
        ix.idf = idf
        ix.avgdl = float(lens.to(torch.float64).mean().to(torch.float32).item())
    if n_chunks is None:
        n_chunks = 5 * N
    if n_chunks:
        cnt = (1 + torch.poisson(torch.full((N,), 4.0, device=dev), generator=g)).clamp_(1, 64).to(torch.int64)
        diff = int(n_chunks) - int(cnt.sum().item())
        guard = 0
        while diff != 0 and guard < 64:
            guard += 1
            ok = torch.nonzero((cnt < 64) if diff > 0 else (cnt > 1)).flatten()
            take = min(abs(diff), len(ok))
            sel = ok[torch.randperm(len(ok), generator=g, device=dev)[:take]]
            cnt[sel] += 1 if diff > 0 else -1
            diff = int(n_chunks) - int(cnt.sum().item())
        if diff != 0:
            raise ValueError("cannot reach the requested chunk total with 1..64 chunks per document")
        doc_off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        doc_off[1:] = torch.cumsum(cnt, 0)
        ix.doc_off = doc_off.to(torch.int32)
        C = int(n_chunks)
        ix.chunk_ids = torch.arange(C, device=dev, dtype=torch.int64)          # contiguous per doc, doc order
        if with_embeddings:
            emb = torch.empty((C, DIM), dtype=torch.float32, device=dev)
            for s in range(0, C, emb_block):
                x = torch.randn((min(emb_block, C - s), DIM), generator=g, device=dev)
                emb[s:s + len(x)] = x / x.norm(dim=1, keepdim=True)
            ix.emb = emb
    return ix


def synthetic_shard(n_docs, n_chunks, n_terms, rank, world, seed=SEED, device="cpu", city_frac=0.85,
                    mean_log_len=5.6, sigma_len=0.8, max_len=20000, block_docs=1 << 16):
    """Shard `rank` of `world` of a corpus with the statistics of synthetic_corpus, WITHOUT materialising the whole
    postings on every rank (bench.py, N > 1): the token stream is generated in blocks of `block_docs` documents, each
    from its own seeded generator, so every rank can produce every block; a rank counts document frequencies over
    ALL blocks (idf, avgdl and the query pool must be the global ones) but keeps postings only for its own document
    range.  The range comes from the chunk-count-balanced cut of CorpusIndex.shard_bounds.  Embeddings are left to
    the caller (i.i.d. rows: a shard-local stream has the same distribution).  Returns (shard, df) with df the GLOBAL
    document frequencies (int64 [V])."""
    dev = torch.device(device)
    g = _gen(dev, seed)
    N = int(n_docs)
    V = max(int(n_terms or (200_000 if N <= 200_000 else 1_000_000)), 64)
    gaps = torch.randint(1, 5, (N,), generator=g, device=dev, dtype=torch.int64)
    doc_ids = torch.cumsum(gaps, 0) + 1000
    lens = torch.exp(mean_log_len + sigma_len * torch.randn(N, generator=g, device=dev)).clamp_(8, max_len).to(torch.int64)
    forced = torch.rand(N, generator=g, device=dev) < city_frac
    full = CorpusIndex(doc_ids=doc_ids, total_docs=N, n_docs_global=N)
    if n_chunks:
        cnt = (1 + torch.poisson(torch.full((N,), 4.0, device=dev), generator=g)).clamp_(1, 64).to(torch.int64)
        diff = int(n_chunks) - int(cnt.sum().item())
        guard = 0
        while diff != 0 and guard < 64:
            guard += 1
            ok = torch.nonzero((cnt < 64) if diff > 0 else (cnt > 1)).flatten()
            sel = ok[torch.randperm(len(ok), generator=g, device=dev)[:min(abs(diff), len(ok))]]
            cnt[sel] += 1 if diff > 0 else -1
            diff = int(n_chunks) - int(cnt.sum().item())
        if diff != 0:
            raise ValueError("cannot reach the requested chunk total with 1..64 chunks per document")
        doc_off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        doc_off[1:] = torch.cumsum(cnt, 0)
        full.doc_off = doc_off.to(torch.int32)
    b = full.shard_bounds(world)
    d0, d1 = int(b[rank]), int(b[rank + 1])
    w = 1.0 / torch.arange(1, V, device=dev, dtype=torch.float64) ** 1.07
    cdf = torch.cumsum(w, 0)
    cdf = (cdf / cdf[-1]).to(torch.float32)
    df = torch.zeros(V, dtype=torch.int64, device=dev)
    kept = []
    for blk, s0 in enumerate(range(0, N, block_docs)):
        s1 = min(N, s0 + block_docs)
        gb = _gen(dev, seed + 104729 * (blk + 1))
        bl = lens[s0:s1]
        total = int(bl.sum().item())
        doc_of = torch.repeat_interleave(torch.arange(s0, s1, device=dev), bl)
        u = torch.rand(total, generator=gb, device=dev)
        key = doc_of * V + (torch.searchsorted(cdf, u).clamp_(max=V - 2) + 1)
        start = torch.cumsum(bl, 0) - bl
        f = forced[s0:s1]
        key[start[f]] = torch.arange(s0, s1, device=dev)[f] * V               # first token -> term 0
        uk, tf = torch.unique_consecutive(torch.sort(key).values, return_counts=True)
        p_doc, p_term = uk // V, uk % V
        df += torch.bincount(p_term, minlength=V)
        if s1 > d0 and s0 < d1:
            m = (p_doc >= d0) & (p_doc < d1)
            kept.append((p_doc[m] - d0, p_term[m], tf[m]))
    p_doc = torch.cat([k[0] for k in kept]) if kept else torch.zeros(0, dtype=torch.int64, device=dev)
    p_term = torch.cat([k[1] for k in kept]) if kept else torch.zeros(0, dtype=torch.int64, device=dev)
    tf = torch.cat([k[2] for k in kept]) if kept else torch.zeros(0, dtype=torch.int64, device=dev)
    order = torch.argsort(p_term, stable=True)                                # (term, doc): blocks are in doc order
    p_doc, p_term, tf = p_doc[order], p_term[order], tf[order]
    term_off = torch.zeros(V + 1, dtype=torch.int64, device=dev)
    term_off[1:] = torch.cumsum(torch.bincount(p_term, minlength=V), 0)
    dff = df.to(torch.float64)
    idf = torch.log10((N - dff + 0.5) / (dff + 0.5)).to(torch.float32)
    idf[df == 0] = 0.0
    sh = CorpusIndex(doc_ids=doc_ids[d0:d1], total_docs=N, n_docs_global=N, doc_base=d0,
                     doc_len=lens[d0:d1].to(torch.int32), term_off=term_off, post_doc=p_doc.to(torch.int32),
                     post_tf=tf.to(torch.int32), idf=idf,
                     avgdl=float(lens.to(torch.float64).mean().to(torch.float32).item()))
    if n_chunks:
        c0 = int(full.doc_off[d0])
        sh.doc_off = full.doc_off[d0:d1 + 1] - c0
        sh.row_base = c0
        sh.chunk_ids = torch.arange(c0, int(full.doc_off[d1]), device=dev, dtype=torch.int64)
    sh._url_group = (torch.arange(d0, d1, dtype=torch.int32)).numpy()        # every document its own (global) group
    return sh, df


def synthetic_query_terms(df, n_queries, seed=SEED + 1, lo_rank=50, hi_rank=50000):
    """The term lists of synthetic_queries from a document-frequency array alone (the sharded bench path)."""
    df = df.cpu().numpy() if torch.is_tensor(df) else np.asarray(df)
    rng = np.random.default_rng(seed)
    V = len(df)
    cand = np.arange(min(lo_rank, V - 1), min(hi_rank, V))
    cand = cand[df[cand] > 0]
    w = df[cand].astype(np.float64) ** 0.5
    w /= w.sum()
    terms = []
    for _ in range(n_queries):
        k = int(rng.integers(1, 5))
        t = [0] + [int(x) for x in rng.choice(cand, size=min(k, len(cand)), replace=False, p=w)]
        if rng.random() < 0.05:
            t.append(t[-1])
        rng.shuffle(t)
        terms.append(t)
    return terms


def synthetic_queries(ix, n_queries, seed=SEED + 1, device=None, lo_rank=50, hi_rank=50000):
    """-> (terms: list[list[int]] WITH repeats, in query order; qvec: float32 [n, 768] tensor)."""
    dev = torch.device(device) if device is not None else (
        ix.emb.device if torch.is_tensor(ix.emb) else torch.device("cpu"))
    rng = np.random.default_rng(seed)
    terms = []
    if ix.term_off is not None:
        toff = ix.term_off.cpu().numpy() if torch.is_tensor(ix.term_off) else np.asarray(ix.term_off)
        df = np.diff(toff)
        V = len(df)
        lo, hi = min(lo_rank, V - 1), min(hi_rank, V)
        cand = np.arange(lo, hi)
        cand = cand[df[cand] > 0]
        w = df[cand].astype(np.float64) ** 0.5
        w /= w.sum()
        for _ in range(n_queries):
            k = int(rng.integers(1, 5))
            t = [0] + [int(x) for x in rng.choice(cand, size=min(k, len(cand)), replace=False, p=w)]
            if rng.random() < 0.05:
                t.append(t[-1])
            rng.shuffle(t)
            terms.append(t)
    else:
        terms = [[] for _ in range(n_queries)]
    qvec = None
    if ix.emb is not None:
        C = ix.emb.shape[0]
        rows = torch.as_tensor(rng.integers(0, C, size=n_queries), device=dev)
        g = _gen(dev, seed)
        emb = ix.emb if torch.is_tensor(ix.emb) else torch.as_tensor(np.asarray(ix.emb))
        base = emb[rows.to(emb.device)].to(dev)
        noise = torch.randn((n_queries, DIM), generator=g, device=dev)
        noise = noise / noise.norm(dim=1, keepdim=True)
        v = base + 0.5 * noise
        v = v / v.norm(dim=1, keepdim=True)
        scale = torch.as_tensor(rng.uniform(5.0, 15.0, size=(n_queries, 1)), dtype=torch.float32, device=dev)
        qvec = (v * scale).to(torch.float32)
    return terms, qvec
