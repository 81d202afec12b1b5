#!/usr/bin/env python3
"""End-to-end drop-in demo (BASELINE configs[0] shape): answer the reference's queries.txt format with the
two-stage path and write `qnum<TAB>rank<TAB>url<TAB>score` lines, like POST /api/batch_search_file
(search_api.py:331-367).  The reference's crawl database is not available offline, so a small synthetic "crawl"
is generated: documents are bags of words from a vocabulary that contains the query words, indexed with
msretr.index_build (the reference's BM25.build_index semantics) and given random unit-norm chunk embeddings.

    python examples/run_queries_txt.py [queries.txt] [out.txt]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from msretr.index_build import bm25_index_from_tokens  # noqa: E402
from msretr.text import preprocess_query, read_queries_file, simple_tokenize  # noqa: E402

DEFAULT_QUERIES = ["tübingen attractions", "food and drinks", "university tuebingen research programs",
                   "castle hohentubingen history", "botanical garden opening hours"]     # shape of queries.txt:1-5


def synthetic_crawl(n_docs=3000, seed=7, queries=DEFAULT_QUERIES):
    rng = np.random.default_rng(seed)
    words = sorted({w for q in queries for w in simple_tokenize(preprocess_query(q))})
    filler = [f"wort{i}" for i in range(400)]
    vocab = words + filler
    p = 1.0 / np.arange(1, len(filler) + 1) ** 0.8
    p = np.r_[np.full(len(words), 0.0035), 0.95 * p / p.sum()]      # a query word is in ~1/3 of the documents
    p /= p.sum()
    doc_ids = (np.cumsum(rng.integers(1, 4, size=n_docs)) + 10).tolist()
    tokens, urls, titles, texts = [], [], [], []
    for i, d in enumerate(doc_ids):
        n = int(rng.integers(20, 200))
        toks = [vocab[j] for j in rng.choice(len(vocab), size=n, p=p)]
        if rng.random() < 0.85:
            toks[0] = "tübingen"
        tokens.append(toks)
        urls.append(f"https://www.site{int(d) % 53}.de/page/{int(d)}" + ("?ref=1" if d % 19 == 0 else ""))
        titles.append(f"Seite {int(d)}")
        texts.append(" ".join(toks))
    ix = bm25_index_from_tokens(doc_ids, tokens)
    ix.urls, ix.titles, ix.texts = urls, titles, texts
    cnt = rng.integers(1, 9, size=len(doc_ids))
    ix.doc_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    C = int(ix.doc_off[-1])
    ix.chunk_ids = np.arange(C, dtype=np.int64)
    emb = rng.standard_normal((C, 768)).astype(np.float32)
    ix.emb = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    return ix


def fake_encoder(dim=768):
    """Stand-in for the sentence encoder (the reference loads a HF model by name): a seeded hash of the text."""
    def enc(text):
        r = np.random.default_rng(abs(hash(text)) % (2 ** 32))
        return (r.standard_normal(dim) * 3).astype(np.float32)
    return enc


def main():
    from msretr.retriever import Retriever
    qfile = sys.argv[1] if len(sys.argv) > 1 else None
    out = sys.argv[2] if len(sys.argv) > 2 else "batch_search_results.txt"
    queries = read_queries_file(qfile) if qfile else [(str(i + 1), q) for i, q in enumerate(DEFAULT_QUERIES)]
    ix = synthetic_crawl()
    rt = Retriever(embedder=fake_encoder(), indexer=ix, tokenizer=simple_tokenize, max_queries=8, max_k=1000)
    res = rt.batch_search(queries)                  # the reference's `results` list (dicts built on access) ...
    res.write(out)                                  # ... and all formatted lines in one native call
    print(f"{len(queries)} queries -> {len(res)} result lines in {out}")
    rt.engine.close()


if __name__ == "__main__":
    main()
