"""Where the select's final kernel spends its time (diagnostic build: s_memtime stamps of workgroup 0 at the phase boundaries,
read back through msr_debug_sel_final): BM25 top-1000 of 1 M synthetic documents, 256 queries.
    python tools/sel_final_phases.py        # on the GPU box"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msretr import _abi
from msretr.build import build_library
_abi.LIB_PATH = build_library(diag=True)
from msretr.engine import DeviceEngine
from msretr.synthetic import SEED, synthetic_corpus, synthetic_queries
dev = torch.device("cuda", 0)
full = synthetic_corpus(1_000_000, n_chunks=0, n_terms=200_000, seed=SEED, device=dev)
terms, _ = synthetic_queries(full, 256, seed=777, device="cpu")
e = DeviceEngine(full, max_queries=256, max_k=1000, rerank_max_docs=0)
packed = e.pack_queries(terms)
for it in range(3):
    e.bm25_topk(None, k=1000, packed=packed); torch.cuda.synchronize()
    ts = (C.c_ulonglong * 16)()
    e.lib.msr_debug_sel_final.argtypes = [C.c_void_p]; e.lib.msr_debug_sel_final.restype = C.c_int
    assert e.lib.msr_debug_sel_final(ts) == 0
    t = list(ts)
    print(json.dumps({"cnt": t[6], "P": t[7], "ticks": [t[i + 1] - t[i] for i in range(5)], "phases": "state | refine | filter-load | pad+sort | output"}))
