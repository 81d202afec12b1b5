"""ctypes binding of libmsretr.so (include/msretr.h).  No torch types cross this boundary: only integers,
raw device addresses and a stream handle.  There is NO CPU fallback: if the library is missing or a call
fails, an exception is raised."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libmsretr.so")

MSR_ABI_VERSION = 3
MSR_CFG_NO_ROW_COPY = 1
MSR_DIM = 768
MSR_MAX_K = 1024
MSR_RERANK_MAX_CHUNKS = 10


class MsrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmsretr error {code}: {msg}")
        self.code = code


class MsrConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32),
                ("max_queries", C.c_int32), ("max_k", C.c_int32), ("rerank_max_docs", C.c_int32),
                ("scan_layout", C.c_int32), ("scan_variant", C.c_int32), ("flags", C.c_int32)]


class MsrRerankParams(C.Structure):
    _fields_ = [("smoothing", C.c_double), ("max_boost", C.c_double), ("max_decay", C.c_double),
                ("max_chunks", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
_SIGNATURES = {
    "msr_abi_version": (C.c_int, []),
    "msr_create": (C.c_int, [C.POINTER(MsrConfig), C.POINTER(_P)]),
    "msr_destroy": (C.c_int, [_P]),
    "msr_last_error": (C.c_char_p, [_P]),
    "msr_bind_postings": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int64, _P, C.c_float,
                                    C.c_double, C.c_double, _P]),
    "msr_bind_chunks": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P]),
    "msr_bind_doc_meta": (C.c_int, [_P, _P, C.c_int64, _P]),
    "msr_scan_arith": (C.c_int, [_P]),
    "msr_scan_width": (C.c_int, [_P]),
    "msr_dense_path": (C.c_int, [_P]),
    "msr_row_copy_state": (C.c_int, [_P]),
    "msr_row_image_state": (C.c_int, [_P]),
    "msr_owned_bytes": (C.c_int64, [_P]),
    # include/msretr_encoder.h
    "msr_enc_layernorm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "msr_enc_attention": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    "msr_enc_geglu": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "msr_enc_linear": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "msr_enc_mean_pool": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "msr_batch_width": (C.c_int, [_P]),
    "msr_batch_gemm_ok": (C.c_int, [_P]),
    "msr_interleave_rows": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "msr_bm25_topk": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P]),
    "msr_dense_topk": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "msr_dense_split_max": (C.c_int, [_P, C.c_int32]),
    "msr_dense_topk_begin": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "msr_dense_topk_end": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "msr_enable_bf16": (C.c_int, [_P, _P]),
    "msr_dense_topk_bf16": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "msr_rerank": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32, C.POINTER(MsrRerankParams), _P, _P, _P,
                             _P, _P, _P, _P]),
    "msr_bind_doc_domains": (C.c_int, [_P, _P, C.c_int64, _P]),
    "msr_diversify": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_double, C.c_int32, _P, _P, _P, _P,
                                _P, _P]),
    "msr_format_lines": (C.c_int64, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, C.c_int64, C.c_int64, _P, C.c_int64]),
    "msr_rerank_gather": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "msr_rerank_gather_blocks": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32,
                                           C.c_int64, _P]),
    "msr_rerank_fuse": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, C.POINTER(MsrRerankParams), _P, _P,
                                  _P, _P, _P, _P, _P]),
    "msr_rerank_combine": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    "msr_rerank_plan": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "msr_rerank_gather_records": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64,
                                            _P]),
    "msr_rerank_scatter": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "msr_merge_topk": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "msr_merge_topk_payload": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                         _P, _P]),
    "msr_build_postings": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    "msr_set_timing": (C.c_int, [_P, C.c_int32]),
    "msr_tune": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "msr_kernel_time_ms": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """dlopen libmsretr.so and attach the prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsrError(-100, f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                             f"g.build()'` (the product path has no CPU fallback)")
    # One HIP runtime per process: torch ships its own libamdhip64 and owns device memory and streams here, so it must be
    # the one our library's HIP symbols resolve to.  Loaded the other way round (libmsretr.so first pulls in the system
    # runtime, torch then brings its own) the second runtime finds no device: msr_create fails with "no HIP device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol from msretr.h is not exported
        fn.restype, fn.argtypes = res, args
    if lib.msr_abi_version() != MSR_ABI_VERSION:
        raise MsrError(-101, f"ABI version {lib.msr_abi_version()} != {MSR_ABI_VERSION}")
    _lib = lib
    return lib


def check(handle, rc):
    if rc != 0:
        msg = load().msr_last_error(handle)
        raise MsrError(rc, msg.decode("utf-8", "replace") if msg else "?")
