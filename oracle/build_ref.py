"""CPU restatement of the reference's BM25 table build.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/indexer/bm25_indexer.py:
  * per-batch tables `BM25._process_document_batch`                                   :196-243
      text handed to the tokeniser: f"{title or ''} {text or ''}"[:1_000_000], lower-cased, "tuebingen" / "tubingen"
      rewritten to "tübingen" (:216-219); a document without tokens gets no row (:222-223); doc_length = number of
      tokens (:225); freq = occurrences of the term in the document (:226-230); rows in the batch's document order,
      terms in first-occurrence order (:232-240); term_updates[term] = {new_docs, freq_increase} (:239-240)
  * corpus statistics `_update_corpus_stats`                                          :346-369
      total_docs = COUNT(*), avg_doc_length = AVG(doc_length) over bm25_doc_stats, stored in REAL (float32) columns
  * idf `_recalculate_idf_scores`                                                     :130-147
      LOG((total_docs - doc_freq + 0.5) / (doc_freq + 0.5)) evaluated by DuckDB and stored REAL.  DuckDB's LOG is log10;
      that and the REAL rounding live in DuckDB, which is not available here: "parity unpinned" for those two facts
      (tests/golden/make_goldens.py lists them), everything above is pinned by tests/golden/bm25_build.json, the output of
      the reference's own `_process_document_batch`.
`index_from_batches` lays the tables out the way the engine binds them (CSR by term over the dense document index = rank
of doc_id ascending, documents ascending inside a term) -- the checker of msr_build_postings (csrc/msr_build.hip) and of
the product's host builders (index_build.py).
"""
import math
from collections import defaultdict

import numpy as np


def process_document_batch(documents, tokenize):
    """documents: list of (doc_id, title | None, text | None); tokenize: str -> list[str] (spaCy in the reference).
    -> (doc_stats [(doc_id, doc_length)], term_freq [(doc_id, term, freq)], term_updates {term: [new_docs, freq_increase]})."""
    doc_stats, term_freq, term_updates = [], [], {}
    for doc_id, title, text in documents:
        combined = f"{title or ''} {text or ''}"
        combined = combined[:1_000_000]
        combined = combined.lower().replace("tuebingen", "tübingen").replace("tubingen", "tübingen")
        tokens = tokenize(combined)
        if not tokens:
            continue
        counts = defaultdict(int)
        for tok in tokens:
            counts[tok] += 1
        doc_stats.append((doc_id, len(tokens)))
        for term, freq in counts.items():
            term_freq.append((doc_id, term, freq))
            u = term_updates.setdefault(term, [0, 0])
            u[0] += 1
            u[1] += freq
    return doc_stats, term_freq, term_updates


def idf_real(total_docs, doc_freq):
    """idf_score as stored: float64 log10 of the ratio, rounded to REAL; total_docs itself round-trips through a REAL
    column (:361-364 write, :133 read)."""
    n_real = float(np.float32(total_docs))
    return np.float32(math.log10((n_real - doc_freq + 0.5) / (doc_freq + 0.5)))


def index_from_batches(doc_stats, term_freq, vocab=None):
    """The tables of one or more batches -> dict of engine-layout arrays: doc_ids (ascending), doc_len, term_off, post_doc
    (dense document index, ascending inside a term), post_tf, idf (float32), avgdl (float32-rounded), total_docs, vocab
    (term -> id; given, or numbered in the order terms first occur with the documents taken in ascending doc_id)."""
    ids = np.array(sorted(d for d, _ in doc_stats), np.int64)
    if len(set(ids.tolist())) != len(ids):
        raise ValueError("duplicate doc_id")
    rank = {int(d): i for i, d in enumerate(ids.tolist())}
    doc_len = np.zeros(len(ids), np.int32)
    for d, l in doc_stats:
        doc_len[rank[int(d)]] = l
    per_doc = defaultdict(list)                                # doc index -> [(term, freq)] in the rows' order
    for d, term, freq in term_freq:
        per_doc[rank[int(d)]].append((term, freq))
    if vocab is None:
        vocab = {}
        for i in range(len(ids)):
            for term, _ in per_doc.get(i, ()):
                vocab.setdefault(term, len(vocab))
    postings = [[] for _ in range(len(vocab))]
    for i in range(len(ids)):
        for term, freq in per_doc.get(i, ()):
            postings[vocab[term]].append((i, freq))
    term_off = np.zeros(len(vocab) + 1, np.int64)
    term_off[1:] = np.cumsum([len(p) for p in postings])
    P = int(term_off[-1])
    post_doc = np.fromiter((d for p in postings for d, _ in p), np.int32, count=P)
    post_tf = np.fromiter((f for p in postings for _, f in p), np.int32, count=P)
    N = len(ids)
    idf = np.array([idf_real(N, len(p)) for p in postings], np.float32)
    avgdl = float(np.float32(doc_len.astype(np.float64).mean())) if N else 0.0
    return dict(doc_ids=ids, doc_len=doc_len, term_off=term_off, post_doc=post_doc, post_tf=post_tf, idf=idf, avgdl=avgdl,
                total_docs=N, vocab=vocab)


def index_from_tokens(doc_ids, token_lists, vocab=None):
    """Already tokenised documents -> engine-layout tables (the token lists stand for what the tokeniser returned)."""
    it = iter(token_lists)
    docs = [(int(d), None, None) for d in doc_ids]
    stats, tf, _ = process_document_batch(docs, lambda _text: list(next(it)))
    return index_from_batches(stats, tf, vocab)
