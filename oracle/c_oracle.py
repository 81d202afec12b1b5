"""ctypes binding of the C restatement (oracle/orc_bm25.c, oracle/orc_dense.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmsr_oracle.so")
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) and build:
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = C.CDLL(LIB)
        _lib.orc_threads.restype = C.c_int
        _lib.orc_bm25_topk.restype = C.c_int
        _lib.orc_dense_topk.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bm25_topk(ix, uterms, qtf, k, min_score=0.0, k1=1.2, b=0.75):
    lib = load()
    t = np.ascontiguousarray(uterms, np.int32); f = np.ascontiguousarray(qtf, np.int32)
    out_doc = np.empty(k, np.int32); out_score = np.empty(k, np.float64)
    n = lib.orc_bm25_topk(_p(ix["term_off"]), _p(ix["post_doc"]), _p(ix["post_tf"]), _p(ix["doc_len"]), _p(ix["idf"]),
                          C.c_float(float(ix["avgdl"])), C.c_double(k1), C.c_double(b), C.c_int64(len(ix["doc_len"])),
                          C.c_int64(len(ix["term_off"]) - 1), _p(t), _p(f), C.c_int(len(t)), C.c_int(k),
                          C.c_double(min_score), _p(out_doc), _p(out_score))
    if n < 0:
        raise MemoryError
    return out_doc[:n].astype(np.int64), out_score[:n]


def dense_topk(emb, doc_off, q, k, max_chunks=0):
    lib = load()
    emb = np.ascontiguousarray(emb, np.float32); doc_off = np.ascontiguousarray(doc_off, np.int32)
    q = np.ascontiguousarray(q, np.float32)
    out_doc = np.empty(k, np.int32); out_score = np.empty(k, np.float32); out_chunk = np.empty(k, np.int32)
    n = lib.orc_dense_topk(_p(emb), _p(doc_off), C.c_int64(len(doc_off) - 1), _p(q), C.c_int(k), C.c_int(max_chunks),
                           _p(out_doc), _p(out_score), _p(out_chunk))
    if n < 0:
        raise MemoryError
    return out_doc[:n].astype(np.int64), out_score[:n], out_chunk[:n].astype(np.int64)


def threads():
    return load().orc_threads()


def set_threads(n):
    load().orc_set_threads(C.c_int(int(n)))
