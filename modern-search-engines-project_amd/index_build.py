"""BM25 index build from tokenised documents (host-side, offline like the reference's).

Produces the same tables as BM25.build_index (indexer/bm25_indexer.py:16-54, 203-250, 346-369, 130-147) from
already-tokenised documents -- the spaCy lemmatiser stays external:
  * a document with no tokens gets no bm25_doc_stats row (:224, :45-46)
  * doc_length = number of tokens, freq = occurrences of the term in the document (:47-53)
  * total_docs = COUNT(*) of bm25_doc_stats, avg_doc_length = AVG(doc_length), both stored REAL (float32)
  * idf = LOG((N - df + 0.5) / (df + 0.5)) evaluated by DuckDB (LOG = log10) and stored REAL
The result is a CorpusIndex in the engine's layout (CSR by term, documents ascending inside a term).
"""
import math
from collections import Counter

import numpy as np

from .index import CorpusIndex
from .text import CITY


def normalise_document_text(title, text):
    """What the reference feeds to the tokeniser: title + text, lower-cased, city spellings unified, capped at
    1 M characters (bm25_indexer.py:30-32)."""
    s = f"{title or ''} {text or ''}".lower().replace("tuebingen", CITY).replace("tubingen", CITY)
    return s[:1_000_000]


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
    n_real = float(np.float32(N))                  # total_docs round-trips through a REAL column (:361-364, :133)
    idf = np.array([np.float32(math.log10((n_real - len(p) + 0.5) / (len(p) + 0.5))) for p in postings], np.float32)
    avgdl = float(np.float32(doc_len.astype(np.float64).mean())) if N else 0.0
    ix = CorpusIndex(doc_ids=ids, doc_len=doc_len, term_off=term_off, post_doc=post_doc, post_tf=post_tf, idf=idf,
                     avgdl=avgdl, total_docs=N, k1=k1, b=b, vocab=vocab)
    ix.n_docs_global = N
    return ix
