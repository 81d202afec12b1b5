"""BM25 index build from tokenised documents (host-side, offline like the reference's).

Produces the same tables as BM25.build_index (indexer/bm25_indexer.py:16-54, 203-250, 346-369, 130-147) from
already-tokenised documents -- the spaCy lemmatiser stays external:
  * a document with no tokens gets no bm25_doc_stats row (:224, :45-46)
  * doc_length = number of tokens, freq = occurrences of the term in the document (:47-53)
  * total_docs = COUNT(*) of bm25_doc_stats, avg_doc_length = AVG(doc_length), both stored REAL (float32)
  * idf = LOG((N - df + 0.5) / (df + 0.5)) evaluated by DuckDB (LOG = log10) and stored REAL
The result is a CorpusIndex in the engine's layout (CSR by term, documents ascending inside a term).
"""
from collections import Counter

import math

import numpy as np

from .index import CorpusIndex
from .text import CITY


def normalise_document_text(title, text):
    """What the reference feeds to the tokeniser: title + text, lower-cased, city spellings unified, capped at
    1 M characters (bm25_indexer.py:30-32)."""
    s = f"{title or ''} {text or ''}".lower().replace("tuebingen", CITY).replace("tubingen", CITY)
    return s[:1_000_000]


def idf_real(total_docs, doc_freq):
    """idf_score of every term at once: LOG((N - df + 0.5) / (df + 0.5)) (bm25_indexer.py:138; DuckDB's LOG is log10) in
    float64, stored REAL (float32); N itself round-trips through a REAL column (:361-364, :133).  doc_freq: integer array."""
    n_real = float(np.float32(total_docs))
    df = np.asarray(doc_freq, np.float64)
    ratio = (n_real - df + 0.5) / (df + 0.5)
    # the logarithm goes through libm's log10 (what DuckDB's LOG and the reference-executed fixture resolve to): numpy's
    # vectorised log10 differs from it in the last float64 ulp on ~1.6 % of inputs, which could flip a float32 rounding.
    # Evaluated once per DISTINCT document frequency (a Zipfian vocabulary has few of them).
    uniq, inv = np.unique(ratio, return_inverse=True)
    logs = np.fromiter((math.log10(x) for x in uniq.tolist()), np.float64, len(uniq))
    return logs[inv].reshape(ratio.shape).astype(np.float32)


def bm25_index_from_tokens(doc_ids, token_lists, k1=1.2, b=0.75):
    """doc_ids: iterable of int; token_lists: one list of term strings per document."""
    rows = sorted(((int(d), toks) for d, toks in zip(doc_ids, token_lists) if toks), key=lambda r: r[0])
    ids = np.array([d for d, _ in rows], np.int64)
    if len(set(ids.tolist())) != len(ids):
        raise ValueError("duplicate doc_id")
    vocab, postings = {}, []                       # postings[t] = [(doc index, tf)] in ascending doc order
    doc_len = np.zeros(len(rows), np.int32)
    for i, (_, toks) in enumerate(rows):
        doc_len[i] = len(toks)
        for term, tf in Counter(toks).items():
            t = vocab.setdefault(term, len(vocab))
            if t == len(postings):
                postings.append([])
            postings[t].append((i, tf))
    V, N = len(vocab), len(rows)
    term_off = np.zeros(V + 1, np.int64)
    term_off[1:] = np.cumsum([len(p) for p in postings])
    post_doc = np.fromiter((d for p in postings for d, _ in p), np.int32, count=int(term_off[-1]))
    post_tf = np.fromiter((tf for p in postings for _, tf in p), np.int32, count=int(term_off[-1]))
    idf = idf_real(N, np.diff(term_off))
    avgdl = float(np.float32(doc_len.astype(np.float64).mean())) if N else 0.0
    ix = CorpusIndex(doc_ids=ids, doc_len=doc_len, term_off=term_off, post_doc=post_doc, post_tf=post_tf, idf=idf,
                     avgdl=avgdl, total_docs=N, k1=k1, b=b, vocab=vocab)
    ix.n_docs_global = N
    return ix


def bm25_index_from_token_ids(doc_ids, tok_off, tok_ids, n_terms, device="cpu", k1=1.2, b=0.75, vocab=None):
    """The same tables from token-id streams, built on `device` (the build given pre-tokenised
    documents, at corpus scale: one radix sort of (term, document) keys + run lengths instead of Python dicts).

    doc_ids int64 [N]; tok_off int64 [N+1]; tok_ids int32 [T] with document i's tokens at tok_off[i]:tok_off[i+1], term
    ids in [0, n_terms).  Documents without tokens get no row (bm25_indexer.py:224); documents are numbered by
    ascending doc_id; inside a term the postings ascend by document.  On a GPU device the tables come from the hand-written
    kernels of csrc/msr_build.hip (per-document sort + run lengths, stable radix sort by term, boundary-based doc_freq,
    three-kernel scans; msr_build_postings); on the CPU device torch's sort / unique_consecutive / bincount restate the same
    build (host-side reference for the tests).  idf is evaluated on the host with the same float64 log10 -> float32
    rounding as bm25_index_from_tokens so that all builders agree bit for bit."""
    import torch
    dev = torch.device(device)
    if dev.type == "cuda":
        return _bm25_index_from_token_ids_hip(doc_ids, tok_off, tok_ids, n_terms, dev, k1, b, vocab)
    ids = torch.as_tensor(np.asarray(doc_ids, np.int64))
    off = torch.as_tensor(np.asarray(tok_off, np.int64)).to(dev)
    tok = (tok_ids if torch.is_tensor(tok_ids) else torch.as_tensor(np.asarray(tok_ids, np.int32))).to(dev)
    if len(set(ids.tolist())) != len(ids):
        raise ValueError("duplicate doc_id")
    lens = (off[1:] - off[:-1])
    if tok.numel() and (int(tok.min()) < 0 or int(tok.max()) >= n_terms):
        raise ValueError("token id outside [0, n_terms)")
    order = torch.argsort(ids).to(dev)                                      # ascending doc_id
    keep = order[lens[order] > 0]                                           # documents with tokens, in id order
    N = int(keep.numel())
    rank = torch.full((len(ids),), -1, dtype=torch.int64, device=dev)
    rank[keep] = torch.arange(N, device=dev)
    tok_doc = torch.repeat_interleave(rank, lens)                           # dense document index of every token
    key = tok.to(torch.int64) * max(N, 1) + tok_doc                         # (term, document): tokens of dropped docs have none
    key = torch.sort(key).values
    uniq, tf = torch.unique_consecutive(key, return_counts=True)
    p_term, p_doc = uniq // max(N, 1), uniq % max(N, 1)
    df = torch.bincount(p_term, minlength=n_terms)
    term_off = torch.zeros(n_terms + 1, dtype=torch.int64, device=dev)
    term_off[1:] = torch.cumsum(df, 0)
    doc_len = lens[keep].to(torch.int32)
    idf = idf_real(N, df.cpu().numpy())
    avgdl = float(np.float32(doc_len.to(torch.float64).mean().item())) if N else 0.0
    ix = CorpusIndex(doc_ids=ids[keep.cpu()].numpy(), doc_len=doc_len, term_off=term_off, post_doc=p_doc.to(torch.int32),
                     post_tf=tf.to(torch.int32), idf=torch.as_tensor(idf).to(dev), avgdl=avgdl, total_docs=N, k1=k1, b=b,
                     vocab=vocab)
    ix.n_docs_global = N
    return ix


def _bm25_index_from_token_ids_hip(doc_ids, tok_off, tok_ids, n_terms, dev, k1, b, vocab):
    """bm25_index_from_token_ids on the GPU through the C ABI (msr_build_postings); no fallback."""
    import ctypes as C

    import torch

    from . import _abi
    lib = _abi.load()
    ids = np.asarray(doc_ids, np.int64)
    if len(set(ids.tolist())) != len(ids):
        raise ValueError("duplicate doc_id")
    off = np.asarray(tok_off.cpu() if torch.is_tensor(tok_off) else tok_off, np.int64)
    tok = (tok_ids if torch.is_tensor(tok_ids) else torch.as_tensor(np.asarray(tok_ids, np.int32))).to(dev, torch.int32).contiguous()
    if tok.numel() and (int(tok.min()) < 0 or int(tok.max()) >= n_terms):
        raise ValueError("token id outside [0, n_terms)")
    lens = np.diff(off)
    order = np.argsort(ids, kind="stable")
    keep = order[lens[order] > 0]                                           # documents with tokens, ascending doc_id
    N = len(keep)
    # the kernels want the kept documents' tokens back to back in that order: one gather of token ranges (skipped when the
    # input already is in that shape)
    if N == len(ids) and np.array_equal(keep, np.arange(N)):
        k_off, k_tok = off, tok
    else:
        k_off = np.zeros(N + 1, np.int64); k_off[1:] = np.cumsum(lens[keep])
        src = torch.as_tensor(np.repeat(off[keep] - k_off[:-1], lens[keep]) + np.arange(k_off[-1]), device=dev)
        k_tok = tok[src].contiguous()
    d_off = torch.as_tensor(k_off).to(dev)
    term_off = torch.empty(n_terms + 1, dtype=torch.int64, device=dev)
    n_post = C.c_int64(0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)
    with torch.cuda.device(dev):
        _abi.check(None, lib.msr_build_postings(ptr(d_off), ptr(k_tok), N, int(n_terms), ptr(term_off), C.c_void_p(0), C.c_void_p(0), 0,
                                                C.byref(n_post), stream))
        P = int(n_post.value)
        post_doc = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
        post_tf = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
        _abi.check(None, lib.msr_build_postings(ptr(d_off), ptr(k_tok), N, int(n_terms), ptr(term_off), ptr(post_doc), ptr(post_tf), max(P, 1),
                                                C.byref(n_post), stream))
    idf = idf_real(N, np.diff(term_off.cpu().numpy()))
    doc_len = torch.as_tensor(lens[keep].astype(np.int32)).to(dev)
    avgdl = float(np.float32(lens[keep].astype(np.float64).mean())) if N else 0.0
    ix = CorpusIndex(doc_ids=ids[keep], doc_len=doc_len, term_off=term_off, post_doc=post_doc[:P], post_tf=post_tf[:P],
                     idf=torch.as_tensor(idf).to(dev), avgdl=avgdl, total_docs=N, k1=k1, b=b, vocab=vocab)
    ix.n_docs_global = N
    return ix
