"""CPU oracle for the two-stage retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's arithmetic for the path
`BM25.search -> /rerank -> Retriever.quick_search`.  It exists to check the HIP
path; it is never imported by the product package.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import or execute anything in here.

Pinning status: every function below is checked in `tests/test_oracle_golden.py` against the fixtures in
`tests/golden/`, which were produced by executing the reference's own function bodies
(`tests/golden/make_goldens.py`).  The one routine with no reference code to execute is
`dense_ref.quick_search` (retriever.py is missing from the reference snapshot): its
arithmetic (cosine, max-pool) is pinned through `rerank_ref.cosine_f32`, its top-k/tie rule is the
build's definition -> "parity unpinned" for that function's selection semantics.
"""
