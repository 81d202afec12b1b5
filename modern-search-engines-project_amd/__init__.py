"""MI355X-native two-stage retriever: BM25 term-at-a-time -> query x chunk cosine -> per-document
max-pool -> fused top-k, as hand-written gfx950 HIP kernels behind a C ABI (include/msretr.h).

Drop-in surface (same names / arguments / return shapes as the reference's call sites):
    BM25(...).search(query, top_k=1000, min_score=0.0)            indexer/bm25_indexer.py:383
    Reranker(...).rerank(doc_ids, similarities, query_embedding)  reranker/reranker_api.py:336 (POST /rerank)
    Retriever(...).quick_search(query, top_k, return_unique_docs) search_api.py:60,87
"""
from ._abi import MsrError  # noqa: F401
from .index import CorpusIndex  # noqa: F401
from .text import preprocess_query, extract_domain, extract_domain_topic  # noqa: F401


def __getattr__(name):
    # heavier modules (torch, the device engine) load on first use
    import importlib
    table = {"DeviceEngine": ".engine", "BM25": ".bm25", "Reranker": ".reranker", "Retriever": ".retriever",
             "ShardedEngine": ".distributed", "synthetic_corpus": ".synthetic", "synthetic_queries": ".synthetic"}
    if name in table:
        return getattr(importlib.import_module(table[name], __name__), name)
    raise AttributeError(name)
