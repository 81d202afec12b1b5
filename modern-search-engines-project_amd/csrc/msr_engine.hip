// C ABI of libmsretr (see include/msretr.h): engine object, scratch, argument checks, kernel sequencing.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/msretr.h"
#include "msr_internal.h"

struct msr_engine {
    msr_config cfg;
    char err[512];
    // bound index parts (borrowed device pointers)
    Bm25Index bm25;
    bool have_postings = false;
    DenseIndex dense;
    bool have_chunks = false;
    const int32_t* url_group = nullptr;
    int64_t url_group_n = 0;
    const int32_t* doc_domain = nullptr;   // msr_bind_doc_domains (borrowed): domain id per document, -1 = rejected from responses
    int64_t doc_domain_n = 0;
    // engine-owned device memory
    int32_t* chunk_doc = nullptr;
    void* emb_presplit = nullptr;     // scan_variant 15: f16 hi/lo image of the rows
    void* row_meta = nullptr;         // packed {document, inverse norm} per row for the K-split kernels
    float* inv_norm_own = nullptr;
    int32_t* span_doc = nullptr;
    int32_t* wspan_doc = nullptr;
    int32_t* wspan12_doc = nullptr;
    float* qn = nullptr;              // [128][768] normalised queries of the current slice
    float* rr_qn = nullptr;           // [max(max_queries, 128)][768] normalised queries of a rerank gather (one launch per call)
    void* qimg = nullptr;             // query image in fragment order (<= 256 KB)
    void* emb_bf16 = nullptr;         // bf16 copy of the embeddings (msr_enable_bf16)
    void* score_rows = nullptr;       // max_queries rows of n_docs float64 (reused as float32 rows)
    size_t score_rows_bytes = 0;
    int32_t* bm_heavy_id = nullptr;    // skip table of the BM25 stage (see Bm25Index)
    void* bm_post = nullptr;           // {doc, tf, tf_component} copy of the postings (see Bm25Index)
    int32_t* bm_dense_id = nullptr;    // dense tf_component tables of the long negative-idf lists (see Bm25Index)
    double* bm_dense = nullptr;
    uint32_t* bm_tile_off = nullptr;
    int32_t* bm_cand_doc = nullptr;    // max_queries rows of n_docs i32: document of each BM25 candidate
    int32_t* bm_cand_n = nullptr;      // [max_queries][tiles] candidates per (query, segment) of the candidate rows
    uint64_t* bm_win = nullptr;        // [max_queries] anchor of the select's window pass (msr_bm25_window)
    size_t bm_cand_bytes = 0;
    SelScratch sel{};
    float* rerank_cos = nullptr;
    int32_t* rerank_meta = nullptr;
    // batched bf16 path scratch (allocated by msr_enable_bf16)
    int32_t* bt_top_doc = nullptr; float* bt_top_score = nullptr; int32_t* bt_top_n = nullptr;
    int32_t* bt_cand_doc = nullptr; float* bt_cand_score = nullptr; int32_t* bt_cand_chunk = nullptr;
    int32_t* bt_cand_n = nullptr;
    // row tiles of <= 256 rows cut at document boundaries (both GEMM paths); built when the chunks are bound
    int32_t* tile_row = nullptr;
    int n_tiles = 0;
    bool tiles_ok = false;             // every document fits one tile
    // default scan for 65..128 queries as a tiled GEMM over the f32 rows (msr_gemm_f32.hip)
    GemmF32Index gf{};
    bool gf_ok = false;
    float* gf_inv_pad = nullptr; void* gf_qimg = nullptr; float* gf_tmax_t = nullptr; float* gf_tmax = nullptr;
    float* gf_thr = nullptr; float* gf_thr2 = nullptr; int32_t* gf_flag = nullptr; void* gf_wvbuf = nullptr;
    int32_t* gf_wv_count = nullptr; void* gf_pairs = nullptr; int32_t* gf_pair_n = nullptr; int32_t* gf_gate = nullptr;
    uint32_t* gf_err = nullptr; float* gf_margin = nullptr; int32_t* gf_cand_doc = nullptr; float* gf_cand_score = nullptr;
    int32_t* gf_cand_chunk = nullptr; int32_t* gf_cand_n = nullptr; float* gf_qn = nullptr; void* gf_fb_qimg = nullptr;
    void* gf_emb_tiled = nullptr;     // fragment-order copy of the f32 rows (256-query streaming pass)
    void* gf_emb_f16 = nullptr;       // row-major f16 image of the rows (launches of several 256-query groups)
    int32_t* tile_trow = nullptr;     // [n_tiles] first row of each tile in those copies
    int64_t n_trows = 0;
    // batched path as a tiled GEMM (msr_gemm.hip): unit-row bf16 image + tile table + scratch for GM_SLICE queries per pass
    GemmIndex gemm{};
    bool gemm_ok = false;
    void* gm_emb_n = nullptr; void* gm_qmat = nullptr; float* gm_tmax = nullptr; float* gm_tmax_t = nullptr;
    float* gm_thr = nullptr; float* gm_thr2 = nullptr; int32_t* gm_flag = nullptr; void* gm_wgbuf = nullptr;
    int32_t* gm_wv_count = nullptr; void* gm_pairs = nullptr; int32_t* gm_pair_n = nullptr; float* gm_qn = nullptr;
    uint32_t* bf_err = nullptr;        // bits of the largest rounding-error norm of an image row (see msr_batch_margin)
    float* bf_margin = nullptr;        // [GM_SLICE] candidate margin of each query of the current slice
    float* bf_ones = nullptr;          // inverse norms of the unit-row image (all 1) for the <= 128-query bf16 sweeps
    void* bf_row_meta = nullptr;       // {document, 1.0f} per row for the K-split bf16 sweeps
    DenseIndex dense_bf16{};           // `dense` with the unit-row image, its inverse norms and row meta
    int n_cus = 256;
    std::unordered_map<void*, size_t> owned;   // engine-owned device allocations (msr_owned_bytes)
    int split_pending = 0;             // queries of an msr_dense_topk_begin whose msr_dense_topk_end has not come yet
    int row_copy_state = 0;            // fragment-order copy of the rows: 0 not wanted / not applicable, 1 built, 2 declined by
                                       // msr_config.flags, 3 allocation failed (the row-major instantiation of the kernel runs)
    int row_image_state = 0;           // f16 image of the rows (launches of several query groups): the same four states
    int last_dense_width = 0;          // queries per pass over the matrix of the most recent msr_dense_topk call (msr_dense_path)
    // timing
    bool timing = false;
    static constexpr int EV_RING = 256;
    static constexpr int EV_KINDS = 4; // 0 dense scan, 1 BM25 TAAT, 2 GEMM emit pass, 3 GEMM sample pass
    hipEvent_t ev_start[EV_KINDS][EV_RING] = {};
    hipEvent_t ev_stop[EV_KINDS][EV_RING] = {};
    int ev_count[EV_KINDS] = {0, 0, 0, 0};   // launches recorded since msr_set_timing(1)
};

static thread_local char g_create_err[512] = "";

static int fail(msr_engine* e, int code, const char* fmt, ...) {
    char* dst = e ? e->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(e, call)                                                                          \
    do {                                                                                          \
        hipError_t _err = (call);                                                                 \
        if (_err != hipSuccess) return fail(e, MSR_ERR_HIP, "%s: %s", #call, hipGetErrorString(_err)); \
    } while (0)

// Every device allocation of an engine goes through these two: msr_owned_bytes() reports what the handle holds.
static hipError_t eng_malloc(msr_engine* e, void** p, size_t bytes) {
    hipError_t err = hipMalloc(p, bytes);
    if (err == hipSuccess && *p) e->owned[*p] = bytes;
    return err;
}
static void free_dev(msr_engine* e, void* p) {
    if (!p) return;
    e->owned.erase(p);
    (void)hipFree(p);
}

static void free_gemm(msr_engine* e) {
    free_dev(e, e->gm_emb_n); free_dev(e, e->gm_qmat); free_dev(e, e->gm_tmax); free_dev(e, e->gm_tmax_t);
    free_dev(e, e->gm_thr); free_dev(e, e->gm_thr2); free_dev(e, e->gm_flag);
    free_dev(e, e->gm_wgbuf); free_dev(e, e->gm_wv_count); free_dev(e, e->gm_pairs); free_dev(e, e->gm_pair_n); free_dev(e, e->gm_qn);
    free_dev(e, e->bf_ones); free_dev(e, e->bf_row_meta); free_dev(e, e->bf_err); free_dev(e, e->bf_margin);
    e->bf_err = nullptr; e->bf_margin = nullptr;
    e->gm_emb_n = nullptr; e->gm_qmat = nullptr; e->gm_tmax = nullptr; e->gm_tmax_t = nullptr;
    e->gm_thr = e->gm_thr2 = nullptr; e->gm_flag = nullptr;
    e->gm_wgbuf = nullptr; e->gm_wv_count = nullptr; e->gm_pairs = nullptr; e->gm_pair_n = nullptr; e->gm_qn = nullptr;
    e->bf_ones = nullptr; e->bf_row_meta = nullptr;
    e->gemm_ok = false;
}

static void free_gf(msr_engine* e) {
    free_dev(e, e->tile_row); free_dev(e, e->tile_trow); e->tile_trow = nullptr; e->n_trows = 0; free_dev(e, e->gf_inv_pad); free_dev(e, e->gf_qimg); free_dev(e, e->gf_tmax_t); free_dev(e, e->gf_tmax);
    free_dev(e, e->gf_thr); free_dev(e, e->gf_thr2);
    free_dev(e, e->gf_flag); free_dev(e, e->gf_wvbuf); free_dev(e, e->gf_wv_count); free_dev(e, e->gf_pairs); free_dev(e, e->gf_pair_n);
    free_dev(e, e->gf_gate); free_dev(e, e->gf_err); free_dev(e, e->gf_margin); free_dev(e, e->gf_cand_doc); free_dev(e, e->gf_cand_score);
    free_dev(e, e->gf_cand_chunk); free_dev(e, e->gf_cand_n); free_dev(e, e->gf_qn); e->gf_qn = nullptr;
    free_dev(e, e->gf_fb_qimg); e->gf_fb_qimg = nullptr;
    free_dev(e, e->gf_emb_tiled); e->gf_emb_tiled = nullptr;
    free_dev(e, e->gf_emb_f16); e->gf_emb_f16 = nullptr;
    e->gf_err = nullptr; e->gf_margin = nullptr; e->gf_cand_doc = nullptr; e->gf_cand_score = nullptr; e->gf_cand_chunk = nullptr;
    e->gf_cand_n = nullptr;
    e->tile_row = nullptr; e->gf_inv_pad = nullptr; e->gf_qimg = nullptr; e->gf_tmax_t = nullptr; e->gf_tmax = nullptr;
    e->gf_thr = e->gf_thr2 = nullptr;
    e->gf_flag = nullptr; e->gf_wvbuf = nullptr; e->gf_wv_count = nullptr; e->gf_pairs = nullptr; e->gf_pair_n = nullptr;
    e->gf_gate = nullptr;
    e->n_tiles = 0; e->tiles_ok = false; e->gf_ok = false;
}

extern "C" int msr_abi_version(void) { return MSR_ABI_VERSION; }

extern "C" const char* msr_last_error(const msr_engine* e) { return e ? e->err : g_create_err; }

// error text of the handle-less entry points (msr_encoder.hip): read back with msr_last_error(NULL)
int msr_fail_global(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_create_err, sizeof(g_create_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int msr_create(const msr_config* cfg, msr_engine** out) {
    if (!cfg || !out) return fail(nullptr, MSR_ERR_INVALID, "msr_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(msr_config))
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: struct_size %d != %zu", cfg->struct_size, sizeof(msr_config));
    if (cfg->dim != MSR_DIM) return fail(nullptr, MSR_ERR_INVALID, "msr_create: dim must be %d", MSR_DIM);
    if (cfg->max_queries < 1 || cfg->max_queries > 4096)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: max_queries out of range [1, 4096]");
    if (cfg->max_k < 1 || cfg->max_k > MSR_MAX_K)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: max_k out of range [1, %d]", MSR_MAX_K);
    if (cfg->rerank_max_docs < 0 || cfg->rerank_max_docs > 1024)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: rerank_max_docs out of range [0, 1024]");
    if (cfg->scan_layout != 0 && cfg->scan_layout != 1)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: scan_layout must be 0 or 1");
    if (cfg->scan_variant != 0 && cfg->scan_variant != 2 && cfg->scan_variant != 7 && cfg->scan_variant != 14 &&
        cfg->scan_variant != 15)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: scan_variant must be 0, 2, 7, 14 or 15");
    if (cfg->flags & ~(int32_t)MSR_CFG_NO_ROW_COPY)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: unknown flag bits 0x%x", cfg->flags);
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0)
        return fail(nullptr, MSR_ERR_HIP, "msr_create: no HIP device available (%s)",
                    herr == hipSuccess ? "device count is 0" : hipGetErrorString(herr));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MSR_ERR_INVALID, "msr_create: device %d out of range (have %d)", cfg->device, ndev);
    msr_engine* e = new (std::nothrow) msr_engine();
    if (!e) return fail(nullptr, MSR_ERR_NOMEM, "msr_create: out of host memory");
    e->cfg = *cfg;
    e->err[0] = 0;
    memset(&e->bm25, 0, sizeof(e->bm25));
    memset(&e->dense, 0, sizeof(e->dense));
    auto bail = [&](int code, const char* what, hipError_t he) {
        fail(nullptr, code, "msr_create: %s: %s", what, hipGetErrorString(he));
        msr_destroy(e);
        return code;
    };
    if ((herr = hipSetDevice(cfg->device)) != hipSuccess) return bail(MSR_ERR_HIP, "hipSetDevice", herr);
    hipDeviceProp_t prop;
    if ((herr = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return bail(MSR_ERR_HIP, "hipGetDeviceProperties", herr);
    e->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // select scratch and query buffers cover the widest sweep (128 queries) whatever max_queries says
    const size_t nq = (size_t)std::max(cfg->max_queries, 128);
    if ((herr = eng_malloc(e, (void**)&e->qn, 128 * MSR_DIM * sizeof(float))) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc qn", herr);
    if ((herr = eng_malloc(e, &e->qimg, 256 * 1024)) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc qimg", herr);
    if (cfg->rerank_max_docs > 0 && (herr = eng_malloc(e, (void**)&e->rr_qn, nq * MSR_DIM * sizeof(float))) != hipSuccess)
        return bail(MSR_ERR_NOMEM, "hipMalloc rr_qn", herr);
    if ((herr = eng_malloc(e, (void**)&e->sel.hist, nq * MSR_SEL_BINS * sizeof(uint32_t))) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc hist", herr);
    if ((herr = eng_malloc(e, (void**)&e->sel.state, nq * sizeof(SelState))) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc state", herr);
    if ((herr = eng_malloc(e, (void**)&e->sel.cand_hi, nq * MSR_SEL_CAP * sizeof(uint64_t))) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc cand_hi", herr);
    if ((herr = eng_malloc(e, (void**)&e->sel.cand_lo, nq * MSR_SEL_CAP * sizeof(uint32_t))) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc cand_lo", herr);
    if ((herr = eng_malloc(e, (void**)&e->sel.cand_n, nq * sizeof(int32_t))) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc cand_n", herr);
    if ((herr = hipMemset(e->sel.hist, 0, nq * MSR_SEL_BINS * sizeof(uint32_t))) != hipSuccess) return bail(MSR_ERR_HIP, "hipMemset hist", herr);
    if ((herr = hipMemset(e->sel.cand_n, 0, nq * sizeof(int32_t))) != hipSuccess) return bail(MSR_ERR_HIP, "hipMemset cand_n", herr);
    if (cfg->rerank_max_docs > 0) {
        const size_t bytes = nq * (size_t)cfg->rerank_max_docs * MSR_RERANK_MAX_CHUNKS * sizeof(float);
        if ((herr = eng_malloc(e, (void**)&e->rerank_cos, bytes)) != hipSuccess) return bail(MSR_ERR_NOMEM, "hipMalloc rerank_cos", herr);
        if ((herr = eng_malloc(e, (void**)&e->rerank_meta, nq * (size_t)cfg->rerank_max_docs * 3 * sizeof(int32_t))) != hipSuccess)
            return bail(MSR_ERR_NOMEM, "hipMalloc rerank_meta", herr);
    }
    for (int w = 0; w < msr_engine::EV_KINDS; ++w)
        for (int j = 0; j < msr_engine::EV_RING; ++j) {
            if ((herr = hipEventCreate(&e->ev_start[w][j])) != hipSuccess) return bail(MSR_ERR_HIP, "hipEventCreate", herr);
            if ((herr = hipEventCreate(&e->ev_stop[w][j])) != hipSuccess) return bail(MSR_ERR_HIP, "hipEventCreate", herr);
        }
    *out = e;
    return MSR_OK;
}

extern "C" int msr_destroy(msr_engine* e) {
    if (!e) return MSR_OK;
    free_dev(e, e->chunk_doc); free_dev(e, e->emb_presplit); free_dev(e, e->row_meta); free_dev(e, e->inv_norm_own); free_dev(e, e->span_doc); free_dev(e, e->wspan_doc); free_dev(e, e->wspan12_doc); free_dev(e, e->qn); free_dev(e, e->rr_qn); free_dev(e, e->qimg); free_dev(e, e->emb_bf16);
    free_dev(e, e->score_rows); free_dev(e, e->bm_heavy_id); free_dev(e, e->bm_post); free_dev(e, e->bm_dense_id); free_dev(e, e->bm_dense); free_dev(e, e->bm_tile_off); free_dev(e, e->bm_cand_doc); free_dev(e, e->bm_cand_n); free_dev(e, e->bm_win); free_dev(e, e->sel.hist); free_dev(e, e->sel.state); free_dev(e, e->sel.cand_hi);
    free_dev(e, e->sel.cand_lo); free_dev(e, e->sel.cand_n); free_dev(e, e->rerank_cos); free_dev(e, e->rerank_meta);
    free_dev(e, e->bt_top_doc); free_dev(e, e->bt_top_score); free_dev(e, e->bt_top_n); free_dev(e, e->bt_cand_doc);
    free_dev(e, e->bt_cand_score); free_dev(e, e->bt_cand_chunk); free_dev(e, e->bt_cand_n);
    free_gemm(e);
    free_gf(e);
    for (int w = 0; w < msr_engine::EV_KINDS; ++w)
        for (int j = 0; j < msr_engine::EV_RING; ++j) {
            if (e->ev_start[w][j]) (void)hipEventDestroy(e->ev_start[w][j]);
            if (e->ev_stop[w][j]) (void)hipEventDestroy(e->ev_stop[w][j]);
        }
    delete e;
    return MSR_OK;
}

// (Re)size the per-slice score rows: max_queries rows of n_docs float64.
static int ensure_score_rows(msr_engine* e, int64_t n_docs) {
    // f64 candidate scores of max_queries queries, or f32 score rows (padded to 32 documents) of up to 128 queries
    const size_t pad = (size_t)(n_docs + 31) / 32 * 32;
    const size_t need = std::max((size_t)e->cfg.max_queries * pad * sizeof(double), (size_t)128 * pad * sizeof(float));
    if (need <= e->score_rows_bytes) return MSR_OK;
    free_dev(e, e->score_rows);
    e->score_rows = nullptr;
    e->score_rows_bytes = 0;
    hipError_t herr = eng_malloc(e, &e->score_rows, need);
    if (herr != hipSuccess) return fail(e, MSR_ERR_NOMEM, "score rows (%zu bytes): %s", need, hipGetErrorString(herr));
    e->score_rows_bytes = need;
    return MSR_OK;
}

extern "C" int msr_bind_postings(msr_engine* e, const int64_t* term_off, int64_t n_terms, const int32_t* post_doc,
                                 const int32_t* post_tf, int64_t n_postings, const int32_t* doc_len, int64_t n_docs,
                                 const float* idf, float avgdl, double k1, double b, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!term_off || !doc_len || !idf || n_terms < 0 || n_postings < 0 || n_docs <= 0 || n_docs >= (1ll << 31) ||
        (n_postings > 0 && (!post_doc || !post_tf)))
        return fail(e, MSR_ERR_INVALID, "msr_bind_postings: bad argument");
    if (e->have_chunks && e->dense.n_docs != n_docs)
        return fail(e, MSR_ERR_INVALID, "msr_bind_postings: n_docs %lld differs from bound chunks (%lld)",
                    (long long)n_docs, (long long)e->dense.n_docs);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // from here on the previous binding's tables are being replaced: the engine counts as unbound until this call succeeds
    // (a failed re-bind must not leave msr_bm25_topk reading freed tables)
    e->have_postings = false;
    int rc = ensure_score_rows(e, n_docs);
    if (rc) return rc;
    {   // candidate lists of the BM25 stage: worst case every document of every query
        const size_t need = (size_t)e->cfg.max_queries * (size_t)n_docs * sizeof(int32_t);
        hipError_t herr;
        if (need > e->bm_cand_bytes) {
            free_dev(e, e->bm_cand_doc); e->bm_cand_doc = nullptr; e->bm_cand_bytes = 0;
            if ((herr = eng_malloc(e, (void**)&e->bm_cand_doc, need)) != hipSuccess)
                return fail(e, MSR_ERR_NOMEM, "BM25 candidate lists (%zu bytes): %s", need, hipGetErrorString(herr));
            e->bm_cand_bytes = need;
        }
        free_dev(e, e->bm_cand_n); e->bm_cand_n = nullptr;
        free_dev(e, e->bm_win); e->bm_win = nullptr;
        if ((herr = eng_malloc(e, (void**)&e->bm_win, (size_t)e->cfg.max_queries * sizeof(uint64_t))) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "msr_bind_postings: %s", hipGetErrorString(herr));
        if ((herr = eng_malloc(e, (void**)&e->bm_cand_n, (size_t)e->cfg.max_queries * msr_bm25_max_segments(n_docs) * sizeof(int32_t))) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "BM25 candidate counts: %s", hipGetErrorString(herr));

    }
    Bm25Index cand{term_off, post_doc, post_tf, doc_len, idf, n_terms, n_postings, n_docs, (double)avgdl, k1, b,
                   nullptr, nullptr, (int32_t)((n_docs + MSR_BM25_TILE - 1) / MSR_BM25_TILE), nullptr, nullptr, nullptr, 0};
    // the scoring kernel indexes LDS with (post_doc - tile start): validate the CSR once, on the device
    hipStream_t st = (hipStream_t)stream;
    int32_t h_flag = 0;
    HIP_TRY(e, msr_bm25_validate(cand, e->sel.cand_n, st));           // cand_n[0] as a scratch word (zero between calls)
    HIP_TRY(e, hipMemcpyAsync(&h_flag, e->sel.cand_n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipMemsetAsync(e->sel.cand_n, 0, sizeof(int32_t), st));
    HIP_TRY(e, hipStreamSynchronize(st));
    if (h_flag >= 1 && h_flag <= 5) {
        static const char* why[] = {"", "term_off is not a monotone offset array ending at n_postings",
                                    "a posting's document index is outside [0, n_docs)",
                                    "documents are not strictly ascending inside a posting list",
                                    "negative doc_len", "non-positive term frequency"};
        return fail(e, MSR_ERR_INVALID, "msr_bind_postings: malformed index: %s", why[h_flag]);
    }
    if (!(avgdl > 0.0f) || !(k1 >= 0.0) || !(b >= 0.0 && b <= 1.0))
        return fail(e, MSR_ERR_INVALID, "msr_bind_postings: avgdl must be > 0, k1 >= 0, 0 <= b <= 1");
    // skip table for the long posting lists (one-time; the offsets come to the host once for this)
    free_dev(e, e->bm_heavy_id); e->bm_heavy_id = nullptr;
    free_dev(e, e->bm_tile_off); e->bm_tile_off = nullptr;
    free_dev(e, e->bm_dense_id); e->bm_dense_id = nullptr;
    free_dev(e, e->bm_dense); e->bm_dense = nullptr;
    std::vector<int32_t> dense_terms;                         // long lists with negative idf, longest first (tables below)
    if (n_terms > 0) {
        std::vector<int64_t> h_toff((size_t)n_terms + 1);
        std::vector<float> h_idf((size_t)n_terms);
        HIP_TRY(e, hipMemcpyAsync(h_toff.data(), term_off, h_toff.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipMemcpyAsync(h_idf.data(), idf, h_idf.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
        std::vector<int32_t> heavy_id((size_t)n_terms, -1), heavy_terms;
        for (int64_t t = 0; t < n_terms; ++t)
            if (h_toff[t + 1] - h_toff[t] >= MSR_BM25_HEAVY_DF && h_toff[t + 1] - h_toff[t] < (1ll << 32)) {
                heavy_id[t] = (int32_t)heavy_terms.size();
                heavy_terms.push_back((int32_t)t);
                if (h_idf[t] < 0.0f) dense_terms.push_back((int32_t)t);
            }
        if (!heavy_terms.empty()) {
            hipError_t herr;
            int32_t* d_terms = nullptr;
            const size_t rows = heavy_terms.size() * (size_t)(cand.n_tiles + 1);
            if ((herr = eng_malloc(e, (void**)&e->bm_heavy_id, heavy_id.size() * sizeof(int32_t))) != hipSuccess ||
                (herr = eng_malloc(e, (void**)&e->bm_tile_off, rows * sizeof(uint32_t))) != hipSuccess ||
                (herr = eng_malloc(e, (void**)&d_terms, heavy_terms.size() * sizeof(int32_t))) != hipSuccess) {
                free_dev(e, d_terms);
                return fail(e, MSR_ERR_NOMEM, "BM25 skip table: %s", hipGetErrorString(herr));
            }
            hipError_t h1 = hipMemcpyAsync(e->bm_heavy_id, heavy_id.data(), heavy_id.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
            if (h1 == hipSuccess) h1 = hipMemcpyAsync(d_terms, heavy_terms.data(), heavy_terms.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
            if (h1 == hipSuccess) h1 = msr_bm25_build_skip(cand, d_terms, (int)heavy_terms.size(), e->bm_tile_off, st);
            const hipError_t h2 = hipStreamSynchronize(st);
            free_dev(e, d_terms);                             // (on every path)
            if (h1 != hipSuccess || h2 != hipSuccess)
                return fail(e, MSR_ERR_HIP, "BM25 skip table: %s", hipGetErrorString(h1 != hipSuccess ? h1 : h2));
            cand.heavy_id = e->bm_heavy_id;
            cand.tile_off = e->bm_tile_off;
        }
        // the longest negative-idf lists get a dense table: at most MSR_BM25_MAX_DENSE of them and 4 GiB in all
        std::stable_sort(dense_terms.begin(), dense_terms.end(), [&](int32_t a, int32_t b2) {
            return h_toff[a + 1] - h_toff[a] > h_toff[b2 + 1] - h_toff[b2];
        });
        const size_t row_bytes = ((size_t)cand.n_tiles * MSR_BM25_TILE + 8) * sizeof(double);
        const size_t cap = std::min<size_t>(MSR_BM25_MAX_DENSE, (size_t)(4ull << 30) / row_bytes);
        if (dense_terms.size() > cap) dense_terms.resize(cap);
    }
    // the copy the scoring kernel streams (after the validation above: the copy is of a well-formed index): every posting with
    // its tf_component, from the per-document length norms k1 (1 - b + b dl / avgdl)
    free_dev(e, e->bm_post); e->bm_post = nullptr;
    {
        const int64_t n_pad = (int64_t)cand.n_tiles * MSR_BM25_TILE;
        double* dnorm = nullptr;
        const size_t bytes = (size_t)(n_postings + 1) * sizeof(Bm25Post);      // + the sentinel posting
        hipError_t herr = eng_malloc(e, &e->bm_post, bytes);
        if (herr != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "postings with tf components (%zu bytes): %s", bytes, hipGetErrorString(herr));
        if ((herr = eng_malloc(e, (void**)&dnorm, (size_t)n_pad * sizeof(double))) != hipSuccess) {
            free_dev(e, e->bm_post); e->bm_post = nullptr;
            return fail(e, MSR_ERR_NOMEM, "length norms: %s", hipGetErrorString(herr));
        }
        hipError_t h1 = msr_bm25_dnorm(doc_len, n_docs, n_pad, k1, b, (double)avgdl, dnorm, st);
        hipError_t h2 = h1 == hipSuccess ? msr_bm25_post_comp(post_doc, post_tf, dnorm, k1, n_postings, (Bm25Post*)e->bm_post, st) : h1;
        hipError_t h3 = hipStreamSynchronize(st);
        free_dev(e, dnorm);
        if (h2 != hipSuccess || h3 != hipSuccess)
            return fail(e, MSR_ERR_HIP, "tf components: %s", hipGetErrorString(h2 != hipSuccess ? h2 : h3));
    }
    cand.post = (const Bm25Post*)e->bm_post;
    if (!dense_terms.empty()) {
        const int64_t stride = (int64_t)cand.n_tiles * MSR_BM25_TILE + 8;      // the tail of a row stays 0.0 (the kernel's "no value")
        std::vector<int32_t> dense_id((size_t)n_terms, -1);
        for (size_t h = 0; h < dense_terms.size(); ++h) dense_id[dense_terms[h]] = (int32_t)h;
        hipError_t herr;
        int32_t* d_terms = nullptr;
        if ((herr = eng_malloc(e, (void**)&e->bm_dense_id, dense_id.size() * sizeof(int32_t))) != hipSuccess ||
            (herr = eng_malloc(e, (void**)&e->bm_dense, dense_terms.size() * (size_t)stride * sizeof(double))) != hipSuccess ||
            (herr = eng_malloc(e, (void**)&d_terms, dense_terms.size() * sizeof(int32_t))) != hipSuccess) {
            free_dev(e, d_terms);
            return fail(e, MSR_ERR_NOMEM, "BM25 dense tables: %s", hipGetErrorString(herr));
        }
        hipError_t h1 = hipMemcpyAsync(e->bm_dense_id, dense_id.data(), dense_id.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
        if (h1 == hipSuccess) h1 = hipMemcpyAsync(d_terms, dense_terms.data(), dense_terms.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
        if (h1 == hipSuccess) h1 = hipMemsetAsync(e->bm_dense, 0, dense_terms.size() * (size_t)stride * sizeof(double), st);
        if (h1 == hipSuccess) h1 = msr_bm25_build_dense(cand, d_terms, (int)dense_terms.size(), e->bm_dense, stride, st);
        const hipError_t h2 = hipStreamSynchronize(st);
        free_dev(e, d_terms);                                 // (on every path)
        if (h1 != hipSuccess || h2 != hipSuccess)
            return fail(e, MSR_ERR_HIP, "BM25 dense tables: %s", hipGetErrorString(h1 != hipSuccess ? h1 : h2));
        cand.dense_id = e->bm_dense_id;
        cand.dense_comp = e->bm_dense;
        cand.dense_stride = stride;
    }
    e->bm25 = cand;
    e->have_postings = true;
    return MSR_OK;
}

extern "C" int msr_bind_chunks(msr_engine* e, const float* emb, int64_t n_chunks, const int32_t* doc_off,
                               int64_t n_docs, const float* inv_norm, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!emb || !doc_off || n_chunks <= 0 || n_chunks >= (1ll << 31) || n_docs <= 0 || n_docs >= (1ll << 31))
        return fail(e, MSR_ERR_INVALID, "msr_bind_chunks: bad argument");
    if (e->have_postings && e->bm25.n_docs != n_docs)
        return fail(e, MSR_ERR_INVALID, "msr_bind_chunks: n_docs %lld differs from bound postings (%lld)",
                    (long long)n_docs, (long long)e->bm25.n_docs);
    if (e->cfg.scan_layout == 1 && !inv_norm)
        return fail(e, MSR_ERR_INVALID, "msr_bind_chunks: inv_norm is required with the interleaved layout");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    int rc = ensure_score_rows(e, n_docs);
    if (rc) return rc;
    // spans need the document offsets on the host (one-time, at bind)
    std::vector<int32_t> h_off((size_t)n_docs + 1);
    HIP_TRY(e, hipMemcpyAsync(h_off.data(), doc_off, (size_t)(n_docs + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    if (h_off[0] != 0 || (int64_t)h_off[n_docs] != n_chunks)
        return fail(e, MSR_ERR_INVALID, "msr_bind_chunks: doc_off[0]=%d, doc_off[n_docs]=%d, n_chunks=%lld",
                    h_off[0], h_off[n_docs], (long long)n_chunks);
    for (int64_t d = 0; d < n_docs; ++d)
        if (h_off[d + 1] < h_off[d]) return fail(e, MSR_ERR_INVALID, "msr_bind_chunks: doc_off not monotone at %lld", (long long)d);
    // equal-chunk-count spans cut at document boundaries: one per workgroup (variant 1) / per wave (variant 2)
    auto make_spans = [&](int target, int64_t min_rows) {
        if ((int64_t)target * min_rows > n_chunks) target = (int)std::max<int64_t>(1, n_chunks / min_rows);
        std::vector<int32_t> sp;
        sp.push_back(0);
        for (int s = 1; s < target; ++s) {
            const int64_t want = n_chunks * s / target;
            int64_t d = std::lower_bound(h_off.begin(), h_off.end(), (int32_t)want) - h_off.begin();   // first doc starting at >= want
            if (d > n_docs) d = n_docs;
            if (d > sp.back()) sp.push_back((int32_t)d);
        }
        if (sp.back() != (int32_t)n_docs) sp.push_back((int32_t)n_docs);
        return sp;
    };
    std::vector<int32_t> spans = make_spans(e->n_cus, 256);
    std::vector<int32_t> wspans = make_spans(e->n_cus * 8, 64);
    std::vector<int32_t> wspans12 = make_spans(e->n_cus * 12, 64);
    // K-split kernels: documents spanned by any two consecutive 16-row groups must fit the LDS ring with a block to spare
    int wide_ok = 1, wide_ok64 = 1;
    {
        const int64_t n_groups = (n_chunks + 15) / 16;
        int64_t dl = 0, dr = 0;                              // document of the window's first / last row
        for (int64_t u = 0; u < n_groups && (wide_ok || wide_ok64); ++u) {
            const int64_t first = 16 * u, last = std::min<int64_t>(16 * (u + 2), n_chunks) - 1;
            while (h_off[dl + 1] <= first) ++dl;
            if (dr < dl) dr = dl;
            while (h_off[dr + 1] <= last) ++dr;
            if (dr - dl + 32 > MSR_WIDE_RING) wide_ok = 0;
            if (dr - dl + 32 > 64) wide_ok64 = 0;
            // ... and one group at most 32 documents: the kernel writes at most two finished blocks per unit
            int64_t dm = dl;
            const int64_t glast = std::min<int64_t>(16 * (u + 1), n_chunks) - 1;
            while (h_off[dm + 1] <= glast) ++dm;
            if (dm - dl > 32) wide_ok = wide_ok64 = 0;
        }
    }
    const int n_spans = (int)spans.size() - 1;
    const int n_wspans = (int)wspans.size() - 1;
    const int n_wspans12 = (int)wspans12.size() - 1;

    free_dev(e, e->chunk_doc); e->chunk_doc = nullptr;
    free_dev(e, e->row_meta); e->row_meta = nullptr;
    free_dev(e, e->emb_presplit); e->emb_presplit = nullptr;
    free_dev(e, e->inv_norm_own); e->inv_norm_own = nullptr;
    free_dev(e, e->span_doc); e->span_doc = nullptr;
    free_dev(e, e->wspan_doc); e->wspan_doc = nullptr;
    free_dev(e, e->wspan12_doc); e->wspan12_doc = nullptr;
    hipError_t herr;
    if ((herr = eng_malloc(e, (void**)&e->chunk_doc, (size_t)n_chunks * sizeof(int32_t))) != hipSuccess)
        return fail(e, MSR_ERR_NOMEM, "chunk_doc: %s", hipGetErrorString(herr));
    if ((herr = eng_malloc(e, (void**)&e->span_doc, spans.size() * sizeof(int32_t))) != hipSuccess)
        return fail(e, MSR_ERR_NOMEM, "span_doc: %s", hipGetErrorString(herr));
    if ((herr = eng_malloc(e, (void**)&e->wspan_doc, wspans.size() * sizeof(int32_t))) != hipSuccess)
        return fail(e, MSR_ERR_NOMEM, "wspan_doc: %s", hipGetErrorString(herr));
    HIP_TRY(e, hipMemcpyAsync(e->span_doc, spans.data(), spans.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(e, hipMemcpyAsync(e->wspan_doc, wspans.data(), wspans.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if ((herr = eng_malloc(e, (void**)&e->wspan12_doc, wspans12.size() * sizeof(int32_t))) != hipSuccess)
        return fail(e, MSR_ERR_NOMEM, "wspan12_doc: %s", hipGetErrorString(herr));
    HIP_TRY(e, hipMemcpyAsync(e->wspan12_doc, wspans12.data(), wspans12.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(e, msr_fill_chunk_doc(doc_off, n_docs, e->chunk_doc, st));
    if (!inv_norm) {
        if ((herr = eng_malloc(e, (void**)&e->inv_norm_own, (size_t)n_chunks * sizeof(float))) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "inv_norm: %s", hipGetErrorString(herr));
        HIP_TRY(e, msr_row_inv_norm(emb, n_chunks, e->inv_norm_own, st));
        inv_norm = e->inv_norm_own;
    }
    if (wide_ok) {
        if ((herr = eng_malloc(e, &e->row_meta, (size_t)(n_chunks + 16) * 8)) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "row_meta: %s", hipGetErrorString(herr));
        HIP_TRY(e, msr_pack_row_meta(e->chunk_doc, inv_norm, n_chunks, e->row_meta, st));
    }
    // The default scan multiplies f16-split pieces (error bound in msr_dense.hip); the bound needs row norms near 1
    // (the reference stores unit-norm rows, indexer/indexer.py:165).  Otherwise fall back to the exact f32 MFMA kernel.
    int variant = e->cfg.scan_variant;
    if (variant == 0) {
        uint32_t h_rng[2] = {0, 0};
        HIP_TRY(e, msr_inv_norm_range(inv_norm, n_chunks, (uint32_t*)e->sel.cand_n, st));   // 2 scratch words
        HIP_TRY(e, hipMemcpyAsync(h_rng, e->sel.cand_n, sizeof(h_rng), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipMemsetAsync(e->sel.cand_n, 0, 2 * sizeof(int32_t), st));
        HIP_TRY(e, hipStreamSynchronize(st));
        float lo, hi;
        memcpy(&lo, &h_rng[0], 4); memcpy(&hi, &h_rng[1], 4);
        variant = (lo >= 0.5f && hi <= 2.0f) ? 7 : 2;
        if (variant == 7 && e->cfg.scan_layout == 0 && wide_ok) variant = 14;   // K-split kernel: up to 64 queries per sweep
    }
    if (variant == 15) {                                  // A/B variant: K-split scan over a pre-split copy of the rows
        if (e->cfg.scan_layout != 0 || !wide_ok) {
            variant = 7;                                  // preconditions of the K-split kernel not met
        } else {
            if ((herr = eng_malloc(e, &e->emb_presplit, (size_t)n_chunks * MSR_DIM * sizeof(float))) != hipSuccess)
                return fail(e, MSR_ERR_NOMEM, "pre-split rows (%zu bytes): %s", (size_t)n_chunks * MSR_DIM * sizeof(float),
                            hipGetErrorString(herr));
            HIP_TRY(e, msr_presplit_rows(emb, n_chunks, e->emb_presplit, st));
        }
    }
    HIP_TRY(e, hipStreamSynchronize(st));                 // spans vector goes out of scope
    e->dense = DenseIndex{emb, doc_off, e->chunk_doc, inv_norm, e->span_doc, n_chunks, n_docs, (n_docs + 31) / 32 * 32, n_spans,
                          e->cfg.scan_layout, e->wspan_doc, n_wspans, e->wspan12_doc, n_wspans12, e->qimg, nullptr,
                          e->emb_presplit, e->row_meta, wide_ok, wide_ok && wide_ok64, nullptr, variant};
    free_dev(e, e->emb_bf16);                                // a new binding invalidates the bf16 copy
    e->emb_bf16 = nullptr;
    free_gemm(e);
    free_gf(e);
    e->row_copy_state = e->row_image_state = 0;              // (a re-bind that does not qualify reports "not applicable")
    // ---- row tiles for the GEMM paths: <= 256 rows, cut at document boundaries (a longer document: no GEMM paths) ----
    std::vector<int32_t> h_trow;                          // first row of each tile in the fragment-order copy (below): every
    int64_t n_trows = 0;                                  // tile starts at a multiple of 16 rows
    {
        std::vector<int32_t> tiles;
        bool ok = true;
        int32_t start = 0;
        tiles.push_back(0);
        for (int64_t d = 0; d < n_docs && ok; ++d) {
            const int32_t end = h_off[d + 1];
            if (end - h_off[d] > 256) ok = false;
            if (end - start > 256) { tiles.push_back(h_off[d]); start = h_off[d]; }
        }
        if (tiles.back() != (int32_t)n_chunks) tiles.push_back((int32_t)n_chunks);
        if (ok) {
            if ((herr = eng_malloc(e, (void**)&e->tile_row, tiles.size() * 4)) != hipSuccess)
                return fail(e, MSR_ERR_NOMEM, "tile table: %s", hipGetErrorString(herr));
            HIP_TRY(e, hipMemcpyAsync(e->tile_row, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(e, hipStreamSynchronize(st));
            e->n_tiles = (int)tiles.size() - 1;
            e->tiles_ok = true;
            h_trow.resize(e->n_tiles);
            for (int t = 0; t < e->n_tiles; ++t) {
                h_trow[t] = (int32_t)n_trows;
                n_trows += (tiles[t + 1] - tiles[t] + 15) / 16 * 16;
            }
            if (n_trows + MSR_STREAM256_TILE_ROWS < ((int64_t)1 << 31)) {
                if ((herr = eng_malloc(e, (void**)&e->tile_trow, (size_t)e->n_tiles * 4)) != hipSuccess)
                    return fail(e, MSR_ERR_NOMEM, "tile table: %s", hipGetErrorString(herr));
                HIP_TRY(e, hipMemcpyAsync(e->tile_trow, h_trow.data(), (size_t)e->n_tiles * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(e, hipStreamSynchronize(st));
                e->n_trows = n_trows;
            }
        }
    }
    // batches of 65..128 queries take ONE streaming pass over the f32 rows (f16 filter + exact f32 finish, msr_gemm_f32.hip)
    // when the corpus allows it
    if (e->tiles_ok && variant == 14 && e->n_tiles >= 64) {
        const int n_tiles = e->n_tiles, nw = e->n_cus * 8, stride = (n_tiles + 31) / 32 * 32;
        // one call holds up to 8 groups of 128 queries (bounded by the select scratch, which covers max(max_queries, 128) rows)
        const int groups = std::min(8, std::max(e->cfg.max_queries, 128) / 128);
        // emitted entries per wave and call: ~600 per 128 queries at 5 M rows (150 x sample stride per query over 2048 waves)
        const int GF_WV_CAP = 4096 * std::max(1, groups / 2);
        const size_t QM = (size_t)groups * 128;
        const int max_nt = std::max(1, std::min(4, groups / 2));      // 256-query groups that share the rows of one launch
        auto alloc = [&](void** p, size_t bytes) { return eng_malloc(e, p, bytes); };
        if ((herr = alloc((void**)&e->gf_inv_pad, (size_t)(n_chunks + 512) * 4)) != hipSuccess ||
            (herr = alloc(&e->gf_qimg, (size_t)groups * 24 * 8192)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_qn, (QM + 64) * MSR_DIM * 4)) != hipSuccess ||
            (herr = alloc(&e->gf_fb_qimg, (QM + 63) / 64 * msr_ksplit_slice_image_bytes())) != hipSuccess ||
            (herr = alloc((void**)&e->gf_tmax_t, (size_t)n_tiles * 8 * (groups >= 2 ? 256 * max_nt : 128) * 4)) != hipSuccess ||   // [tile][wave][queries of a launch]
            (herr = alloc((void**)&e->gf_tmax, QM * stride * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_thr, QM * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_thr2, QM * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_flag, QM * 4)) != hipSuccess ||
            (herr = alloc(&e->gf_wvbuf, (size_t)nw * GF_WV_CAP * 16)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_wv_count, (size_t)nw * 4)) != hipSuccess ||
            (herr = alloc(&e->gf_pairs, QM * 4096 * 8)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_pair_n, QM * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_gate, 16 * 4)) != hipSuccess ||       // one gate word per slice of 64 queries
            (herr = alloc((void**)&e->gf_err, 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_margin, QM * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_cand_doc, QM * MSR_SEL_CAP * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_cand_score, QM * MSR_SEL_CAP * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_cand_chunk, QM * MSR_SEL_CAP * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gf_cand_n, QM * 4)) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "GEMM scan scratch: %s", hipGetErrorString(herr));
        HIP_TRY(e, msr_pad_inv_norm(inv_norm, n_chunks, n_chunks + 512, e->gf_inv_pad, st));
        HIP_TRY(e, hipMemsetAsync(e->gf_pair_n, 0, QM * 4, st));
        HIP_TRY(e, hipMemsetAsync(e->gf_gate, 0, 16 * 4, st));
        HIP_TRY(e, hipMemsetAsync(e->gf_cand_n, 0, QM * 4, st));
        HIP_TRY(e, msr_f16_row_error(emb, inv_norm, n_chunks, e->gf_err, st));   // measured once: the margin of the f16 filter
        // The 256-query kernel streams a copy of the rows in fragment order (whole cache lines per load instruction; +3 % rows
        // of padding: every tile starts at a multiple of 16 rows).  Size of the copy: a workgroup's tile visit ALWAYS loads
        // MSR_STREAM256_TILE_ROWS = 8 waves x 32 rows from the tile's first row on, whatever the tile's own length (rows behind
        // the tile are masked in the epilogue); all K blocks of a visit, and the prefetch of the next visit's first block,
        // address rows of [first row of a tile, first row + MSR_STREAM256_TILE_ROWS) -- the "next" tile of a workgroup's last
        // visit is that same tile again (jn == jt).  So the highest row the kernel touches is max_t tile_trow[t] +
        // MSR_STREAM256_TILE_ROWS - 1 < n_trows + MSR_STREAM256_TILE_ROWS: that many rows of zero padding behind the last
        // tile are exactly enough, for every prefetch depth (checked against the tile table here, not assumed).
        // The row-major matrix stays what every other kernel reads.  Declined by MSR_CFG_NO_ROW_COPY, or when the allocation
        // fails (the copy doubles the matrix): the TILED = false instantiation of the same kernel reads the caller's matrix.
        e->row_copy_state = 0;
        if (groups >= 2 && e->tile_trow) {
            const size_t copy_rows = (size_t)n_trows + MSR_STREAM256_TILE_ROWS;
            if ((int64_t)h_trow.back() + MSR_STREAM256_TILE_ROWS > (int64_t)copy_rows)
                return fail(e, MSR_ERR_INVALID, "msr_bind_chunks: internal: tile table exceeds the fragment-order copy");
            if (e->cfg.flags & MSR_CFG_NO_ROW_COPY) {
                e->row_copy_state = 2;
            } else if ((herr = alloc(&e->gf_emb_tiled, copy_rows * MSR_DIM * 4)) != hipSuccess) {
                (void)hipGetLastError();                      // clear the sticky error: the engine works without the copy
                e->gf_emb_tiled = nullptr;
                e->row_copy_state = 3;
            } else {
                HIP_TRY(e, hipMemsetAsync((char*)e->gf_emb_tiled + (size_t)n_trows * MSR_DIM * 4, 0,
                                          (size_t)MSR_STREAM256_TILE_ROWS * MSR_DIM * 4, st));
                HIP_TRY(e, msr_tile_rows(emb, e->tile_row, e->tile_trow, n_tiles, e->gf_emb_tiled, st));
                e->row_copy_state = 1;
            }
        }
        // Launches of SEVERAL 256-query groups (engines for >= 512 queries per call: the batched steps, a rank of a sharded
        // run) are bound by the matrix pipes and the vector issue beside them; every group converts the same f32 rows to f16
        // again.  They read an f16 image of the rows instead -- the very values the pass converts in registers (round to
        // nearest, not normalised): same products, same candidates, same results; 1536 B per row.  Declined with the other
        // copy (MSR_CFG_NO_ROW_COPY) or when the allocation fails: those launches then convert as before.
        e->row_image_state = 0;
        if (max_nt >= 2 && (e->cfg.flags & MSR_CFG_NO_ROW_COPY)) {
            e->row_image_state = 2;
        } else if (max_nt >= 2) {
            const size_t img_rows = (size_t)n_chunks + 512;
            if ((herr = alloc(&e->gf_emb_f16, img_rows * MSR_DIM * 2)) != hipSuccess) {
                (void)hipGetLastError();
                e->gf_emb_f16 = nullptr;
                e->row_image_state = 3;
            } else {
                HIP_TRY(e, msr_f16_rows(emb, n_chunks, (int64_t)img_rows, e->gf_emb_f16, st));
                e->row_image_state = 1;
            }
        }
        e->gf = GemmF32Index{e->tile_row, n_tiles, e->n_cus, groups, e->gf_inv_pad, e->gf_qimg, e->gf_tmax_t, e->gf_tmax, stride,
                             e->gf_thr, e->gf_thr2, e->gf_flag, e->gf_wvbuf,
                             GF_WV_CAP, e->gf_wv_count, e->gf_pairs, e->gf_pair_n, e->gf_err, e->gf_margin, e->gf_cand_doc,
                             e->gf_cand_score, e->gf_cand_chunk, e->gf_cand_n, max_nt, e->gf_emb_tiled,
                             e->gf_emb_tiled ? e->tile_trow : nullptr, e->gf_emb_f16};
        e->gf_ok = true;
    }
    e->have_chunks = true;
    return MSR_OK;
}

extern "C" int msr_bind_doc_meta(msr_engine* e, const int32_t* url_group, int64_t n_docs, void* stream) {
    (void)stream;
    if (!e) return MSR_ERR_INVALID;
    if (url_group && e->have_chunks && n_docs != e->dense.n_docs)
        return fail(e, MSR_ERR_INVALID, "msr_bind_doc_meta: n_docs mismatch");
    e->url_group = url_group;
    e->url_group_n = n_docs;
    return MSR_OK;
}

extern "C" int msr_bind_doc_domains(msr_engine* e, const int32_t* domain, int64_t n_docs, void* stream) {
    (void)stream;
    if (!e) return MSR_ERR_INVALID;
    if (n_docs < 0 || (domain && n_docs == 0)) return fail(e, MSR_ERR_INVALID, "msr_bind_doc_domains: bad argument");
    e->doc_domain = domain;
    e->doc_domain_n = domain ? n_docs : 0;
    return MSR_OK;
}

extern "C" int msr_diversify(msr_engine* e, int32_t n_queries, const int32_t* fused_doc, const double* fused_score,
                             const double* fused_orig, const int32_t* fused_chunk, const int32_t* fused_n, int32_t max_cand,
                             int32_t top_k, double relevance_threshold, int32_t diversify, int32_t* out_doc, double* out_score,
                             double* out_orig, int32_t* out_chunk, int32_t* out_n, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!fused_doc || !fused_score || !fused_orig || !fused_chunk || !fused_n || !out_doc || !out_score || !out_orig || !out_chunk ||
        !out_n)
        return fail(e, MSR_ERR_INVALID, "msr_diversify: null argument");
    if (n_queries < 0 || max_cand < 1 || max_cand > 1024 || top_k < 1)
        return fail(e, MSR_ERR_INVALID, "msr_diversify: bad argument (max_cand=%d, top_k=%d)", max_cand, top_k);
    if (n_queries == 0) return MSR_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_diversify_run(n_queries, fused_doc, fused_score, fused_orig, fused_chunk, fused_n, max_cand, e->doc_domain,
                                 e->doc_domain_n, top_k, relevance_threshold, diversify, out_doc, out_score, out_orig, out_chunk,
                                 out_n, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_interleave_rows(msr_engine* e, const float* src, int64_t n_rows, float* dst, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!src || !dst || n_rows <= 0 || src == dst) return fail(e, MSR_ERR_INVALID, "msr_interleave_rows: bad argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_interleave(src, n_rows, dst, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_scan_arith(const msr_engine* e) {
    if (!e || !e->have_chunks) return -1;
    const int v = e->dense.variant;
    return (v == 7 || v == 14 || v == 15) ? 1 : 0;
}

extern "C" int msr_scan_width(const msr_engine* e) {
    if (!e || !e->have_chunks) return -1;
    const int v = e->dense.variant;
    const bool wide = (v == 2 || v == 14 || v == 15) && e->dense.layout == 0 && e->dense.wide_ok;
    // streaming pass over the f32 rows for batches of more than 64 queries: 128 queries per pass, 256 for batches of more
    // than 128 (when the engine was created for that many queries per call)
    if (e->gf_ok && wide && v == 14) return e->gf.max_groups >= 2 ? 256 : 128;
    return wide ? 64 : 32;
}

extern "C" int msr_dense_path(const msr_engine* e) { return e ? e->last_dense_width : -1; }

extern "C" int64_t msr_owned_bytes(const msr_engine* e) {
    if (!e) return -1;
    int64_t total = 0;
    for (const auto& kv : e->owned) total += (int64_t)kv.second;
    return total;
}

extern "C" int msr_row_copy_state(const msr_engine* e) { return e ? e->row_copy_state : -1; }
extern "C" int msr_row_image_state(const msr_engine* e) { return e ? e->row_image_state : -1; }

extern "C" int msr_batch_width(const msr_engine* e) {
    if (!e || !e->have_chunks || !e->emb_bf16) return -1;
    return e->dense.wide_ok && e->dense.wide_ok64 ? 128 : 64;
}

extern "C" int msr_set_timing(msr_engine* e, int32_t enabled) {
    if (!e) return MSR_ERR_INVALID;
    e->timing = enabled != 0;
    for (int w = 0; w < msr_engine::EV_KINDS; ++w) e->ev_count[w] = 0;
    return MSR_OK;
}

extern "C" int msr_kernel_time_ms(msr_engine* e, int32_t which, float* out_ms, int32_t* out_launches) {
    if (!e || which < 0 || which >= msr_engine::EV_KINDS || !out_ms) return e ? fail(e, MSR_ERR_INVALID, "msr_kernel_time_ms: bad argument") : MSR_ERR_INVALID;
    const int n = std::min(e->ev_count[which], (int)msr_engine::EV_RING);
    if (n <= 0) return fail(e, MSR_ERR_INVALID, "msr_kernel_time_ms: no timed launch recorded");
    float total = 0.f;
    for (int j = 0; j < n; ++j) {
        float ms = 0.f;
        HIP_TRY(e, hipEventSynchronize(e->ev_stop[which][j]));
        HIP_TRY(e, hipEventElapsedTime(&ms, e->ev_start[which][j], e->ev_stop[which][j]));
        total += ms;
    }
    *out_ms = total;
    if (out_launches) *out_launches = n;
    return MSR_OK;
}

extern "C" int msr_bm25_topk(msr_engine* e, const int32_t* q_term_off, const int32_t* q_terms, const int32_t* q_qtf,
                             int32_t n_queries, int32_t k, double min_score, int32_t* out_doc, double* out_score,
                             int32_t* out_n, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!e->have_postings) return fail(e, MSR_ERR_NOT_BOUND, "msr_bm25_topk: postings not bound");
    if (n_queries < 0 || k < 1 || k > e->cfg.max_k || !q_term_off || !out_doc || !out_score || !out_n)
        return fail(e, MSR_ERR_INVALID, "msr_bm25_topk: bad argument (k=%d, max_k=%d)", k, e->cfg.max_k);
    if (n_queries == 0) return MSR_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int slice = e->cfg.max_queries;
    const int64_t N = e->bm25.n_docs;
    for (int q0 = 0; q0 < n_queries; q0 += slice) {
        const int nq = std::min(slice, n_queries - q0);
        int32_t* o_doc = out_doc + (int64_t)q0 * k;
        double* o_score = out_score + (int64_t)q0 * k;
        // every launch gets its own event pair (ring of EV_RING; later launches are not recorded)
        const bool timed = e->timing && e->ev_count[1] < msr_engine::EV_RING;
        int n_seg = 0;
        int64_t seg_stride = 0;
        if (timed) HIP_TRY(e, hipEventRecord(e->ev_start[1][e->ev_count[1]], st));
        HIP_TRY(e, msr_bm25_scores(e->bm25, q_term_off, q_terms, q_qtf, q0, nq, min_score, (double*)e->score_rows, e->bm_cand_doc,
                                   e->bm_cand_n, &n_seg, &seg_stride, st));
        if (timed) {
            HIP_TRY(e, hipEventRecord(e->ev_stop[1][e->ev_count[1]], st));
            e->ev_count[1]++;
        }
        // scores >= min_score >= 0 lie in a window of 16 octaves below a bound known from the query alone: one histogram
        // pass instead of two (a negative min_score keeps the general two)
        const uint64_t* win = nullptr;
        if (min_score >= 0.0) {
            HIP_TRY(e, msr_bm25_window(e->bm25, q_term_off, q_terms, q_qtf, q0, nq, e->bm_win, st));
            win = e->bm_win;
        }
        HIP_TRY(e, msr_select_topk_list((const double*)e->score_rows, e->bm_cand_doc, e->bm_cand_n, n_seg, seg_stride, N, nq, k,
                                        e->sel, o_doc, o_score, out_n + q0, st, win));
    }
    return MSR_OK;
}

// A query whose entries overflowed in the streaming pass (huge tie groups, a zero vector) raised the gate word of its 64-query
// slice: that slice once more on the sweeps, which handle any input.  One scan per slice, gated on its word, into its own score
// rows; then ONE select and ONE best-chunk pass over all rows of the call, in which a query takes part only if its slice's word
// is up.  When no gate is up (the normal case) all these launches return at once.  Queries: e->gf_qn (normalised, nq of them).
static int dense_gated_fallback(msr_engine* e, int nq, int k, int32_t* out_doc, float* out_score, int32_t* out_chunk,
                                int32_t* out_n, hipStream_t st) {
    const int64_t N = e->dense.n_docs;
    DenseIndex ix = e->dense;
    ix.gate = e->gf_gate;
    HIP_TRY(e, msr_dense_scan_slices(ix, e->gf_qn, nq, (float*)e->score_rows, e->gf_fb_qimg, st));
    SelScratch sel = e->sel;
    sel.gate = e->gf_gate; sel.gate_per64 = 1;
    HIP_TRY(e, msr_select_topk(32, e->score_rows, N, e->dense.score_stride, nq, k, sel, out_doc, out_score, out_n, st));
    ix.gate = e->gf_gate; ix.gate_per64 = 1;
    if (out_chunk) HIP_TRY(e, msr_best_chunk(ix, e->gf_qn, nq, k, 0, out_doc, out_n, out_chunk, st));
    return MSR_OK;
}

static bool dense_stream_ok(const msr_engine* e, int k) {
    const bool wide = (e->dense.variant == 2 || e->dense.variant == 14 || e->dense.variant == 15) && e->dense.layout == 0 &&
                      e->dense.wide_ok;
    return wide && e->gf_ok && e->dense.variant == 14 && e->gf.n_tiles >= 2 * k;
}

extern "C" int msr_dense_split_max(const msr_engine* e, int32_t k) {
    if (!e || !e->have_chunks || k < 1 || k > e->cfg.max_k || !dense_stream_ok(e, k)) return 0;
    return e->gf.max_groups >= 2 ? (e->gf.max_groups & ~1) * 128 : 128;
}

extern "C" int msr_dense_topk_begin(msr_engine* e, const float* q, int32_t n_queries, int32_t k, int32_t k_part, float* out_part,
                                    void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!e->have_chunks) return fail(e, MSR_ERR_NOT_BOUND, "msr_dense_topk_begin: chunks not bound");
    if (e->split_pending) return fail(e, MSR_ERR_INVALID, "msr_dense_topk_begin: the previous begin has not been ended");
    const int cap = msr_dense_split_max(e, k);
    if (!q || !out_part || k_part < 1 || k_part > k || n_queries <= 64 || n_queries > cap)
        return fail(e, MSR_ERR_INVALID, "msr_dense_topk_begin: needs 64 < n_queries <= msr_dense_split_max() = %d, 1 <= k_part <= k (got %d, %d, %d)",
                    cap, n_queries, k_part, k);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_prep_queries(q, n_queries, e->gf_qn, (n_queries + 63) / 64 * 64, st));
    HIP_TRY(e, hipMemsetAsync(e->gf_gate, 0, 16 * 4, st));
    hipEvent_t ev[4];
    const bool timed = e->timing && e->ev_count[0] < msr_engine::EV_RING && e->ev_count[3] < msr_engine::EV_RING;
    if (timed) {
        ev[0] = e->ev_start[3][e->ev_count[3]]; ev[1] = e->ev_stop[3][e->ev_count[3]];
        ev[2] = e->ev_start[0][e->ev_count[0]]; ev[3] = e->ev_stop[0][e->ev_count[0]];
    }
    int width = 0;
    HIP_TRY(e, msr_gemm_f32_pass(e->gf, e->dense, e->gf_qn, n_queries, k, k_part, out_part, timed ? ev : nullptr, &width, st));
    e->last_dense_width = width;
    if (timed) { e->ev_count[0]++; e->ev_count[3]++; }
    e->split_pending = n_queries;
    return MSR_OK;
}

extern "C" int msr_dense_topk_end(msr_engine* e, int32_t n_queries, int32_t k, const float* bound, int32_t* out_doc,
                                  float* out_score, int32_t* out_chunk, int32_t* out_n, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (e->split_pending != n_queries || n_queries <= 0)
        return fail(e, MSR_ERR_INVALID, "msr_dense_topk_end: no matching msr_dense_topk_begin (pending %d, got %d)", e->split_pending, n_queries);
    if (!out_doc || !out_score || !out_n || k < 1 || k > e->cfg.max_k) return fail(e, MSR_ERR_INVALID, "msr_dense_topk_end: bad argument");
    e->split_pending = 0;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_gemm_f32_finish(e->gf, e->dense, e->gf_qn, n_queries, k, bound, out_doc, out_score, out_chunk, out_n, e->gf_gate, st));
    return dense_gated_fallback(e, n_queries, k, out_doc, out_score, out_chunk, out_n, st);
}

extern "C" int msr_dense_topk(msr_engine* e, const float* q, int32_t n_queries, int32_t k, int32_t max_chunks_per_doc,
                              int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!e->have_chunks) return fail(e, MSR_ERR_NOT_BOUND, "msr_dense_topk: chunks not bound");
    if (e->split_pending)
        return fail(e, MSR_ERR_INVALID, "msr_dense_topk: an msr_dense_topk_begin is pending (its scratch is in use): call msr_dense_topk_end first");
    if (n_queries < 0 || k < 1 || k > e->cfg.max_k || max_chunks_per_doc < 0 || !q || !out_doc || !out_score || !out_n)
        return fail(e, MSR_ERR_INVALID, "msr_dense_topk: bad argument (k=%d, max_k=%d)", k, e->cfg.max_k);
    if (n_queries == 0) return MSR_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // one sweep of E serves up to 32 queries (wave-streaming kernel) or 64 (K-split kernel); batches of more than 64
    // queries run as a GEMM over the f32 rows, 128 queries per pass (msr_gemm_f32.hip), when the corpus allows it
    const bool wide = (e->dense.variant == 2 || e->dense.variant == 14 || e->dense.variant == 15) && e->dense.layout == 0 &&
                      e->dense.wide_ok && max_chunks_per_doc == 0;
    const bool gemm = wide && e->gf_ok && e->dense.variant == 14 && e->gf.n_tiles >= 2 * k;
    const int64_t N = e->dense.n_docs;
    e->last_dense_width = gemm && n_queries > 64 ? 0 : (wide ? 64 : 32);       // (the streaming path reports its own width below)
    // sweeps for queries [q0, q0 + cnt): `gate` non-null = fallback launches that only do work when *gate != 0
    auto sweeps = [&](int q0, int cnt, const int32_t* gate) -> int {
        const int slice = wide ? 64 : 32;
        DenseIndex ix = e->dense;
        SelScratch sel = e->sel;
        for (int s0 = q0; s0 < q0 + cnt; s0 += slice) {
            const int nq = std::min(slice, q0 + cnt - s0);
            // gated fallback: one gate word per slice of 64 queries (the streaming path raises the gates of the slices that
            // hold an overflowed query)
            ix.gate = sel.gate = gate ? gate + (s0 - q0) / 64 : nullptr;
            // zero rows up to the query-block count of the kernel that runs (1, 2 or 4 blocks of 16)
            const int nq_pad = nq > 32 ? 64 : (nq > 16 || (wide && e->dense.variant >= 14)) ? 32 : 16;
            const bool timed = !gate && e->timing && e->ev_count[0] < msr_engine::EV_RING;
            HIP_TRY(e, msr_prep_queries(q + (int64_t)s0 * MSR_DIM, nq, e->qn, nq_pad, st));
            if (timed) HIP_TRY(e, hipEventRecord(e->ev_start[0][e->ev_count[0]], st));
            HIP_TRY(e, msr_dense_scan(ix, e->qn, nq, max_chunks_per_doc, (float*)e->score_rows, st));
            if (timed) {
                HIP_TRY(e, hipEventRecord(e->ev_stop[0][e->ev_count[0]], st));
                e->ev_count[0]++;
            }
            HIP_TRY(e, msr_select_topk(32, e->score_rows, N, e->dense.score_stride, nq, k, sel, out_doc + (int64_t)s0 * k,
                                       out_score + (int64_t)s0 * k, out_n + s0, st));
            if (out_chunk)
                HIP_TRY(e, msr_best_chunk(ix, e->qn, nq, k, max_chunks_per_doc, out_doc + (int64_t)s0 * k,
                                          out_n + s0, out_chunk + (int64_t)s0 * k, st));
        }
        return MSR_OK;
    };
    int q0 = 0;
    while (q0 < n_queries) {
        const int left = n_queries - q0;
        if (gemm && left > 64) {
            // more than 128 queries run in groups of 256: with an odd number of 128-query groups only the even part is usable
            const int cap = e->gf.max_groups >= 2 ? (e->gf.max_groups & ~1) * 128 : 128;
            const int nq = std::min(cap, left);
            // (normalised once for the pass AND for the gated sweeps behind it: zero rows up to the last slice's 64)
            HIP_TRY(e, msr_prep_queries(q + (int64_t)q0 * MSR_DIM, nq, e->gf_qn, (nq + 63) / 64 * 64, st));
            HIP_TRY(e, hipMemsetAsync(e->gf_gate, 0, 16 * 4, st));
            hipEvent_t ev[4];
            const bool timed = e->timing && e->ev_count[0] < msr_engine::EV_RING && e->ev_count[3] < msr_engine::EV_RING;
            if (timed) {
                ev[0] = e->ev_start[3][e->ev_count[3]]; ev[1] = e->ev_stop[3][e->ev_count[3]];
                ev[2] = e->ev_start[0][e->ev_count[0]]; ev[3] = e->ev_stop[0][e->ev_count[0]];
            }
            int width = 0;
            HIP_TRY(e, msr_gemm_f32_topk(e->gf, e->dense, e->gf_qn, nq, k, out_doc + (int64_t)q0 * k,
                                         out_score + (int64_t)q0 * k, out_chunk ? out_chunk + (int64_t)q0 * k : nullptr,
                                         out_n + q0, e->gf_gate, timed ? ev : nullptr, &width, st));
            e->last_dense_width = std::max(e->last_dense_width, width);
            if (timed) { e->ev_count[0]++; e->ev_count[3]++; }
            {
                int rcf = dense_gated_fallback(e, nq, k, out_doc + (int64_t)q0 * k, out_score + (int64_t)q0 * k,
                                               out_chunk ? out_chunk + (int64_t)q0 * k : nullptr, out_n + q0, st);
                if (rcf) return rcf;
            }
            q0 += nq;
        } else {
            const int nq = std::min(wide ? 64 : 32, left);
            int rc = sweeps(q0, nq, nullptr);
            if (rc) return rc;
            q0 += nq;
        }
    }
    return MSR_OK;
}

static int rerank_args_ok(msr_engine* e, const char* fn, int32_t n_queries, int32_t max_cand, int32_t max_chunks) {
    if (n_queries < 0 || max_cand < 1 || max_cand > e->cfg.rerank_max_docs)
        return fail(e, MSR_ERR_INVALID, "%s: bad argument (max_cand=%d, rerank_max_docs=%d)", fn, max_cand,
                    e->cfg.rerank_max_docs);
    if (max_chunks < 1 || max_chunks > MSR_RERANK_MAX_CHUNKS)
        return fail(e, MSR_ERR_INVALID, "%s: max_chunks out of range [1, %d]", fn, MSR_RERANK_MAX_CHUNKS);
    return MSR_OK;
}

static constexpr int BT_SLICE = 128;                        // most queries per bf16 K-split sweep
static constexpr int GM_SLICE = 1024;                       // queries per pass of the GEMM path (4 query tiles of 256)
static constexpr int GM_WV_CAP = 16384;                     // emitted entries per wave (x 8 waves x #CU x 16 B = 512 MB at 256 CUs)

extern "C" int msr_enable_bf16(msr_engine* e, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!e->have_chunks) return fail(e, MSR_ERR_NOT_BOUND, "msr_enable_bf16: chunks not bound");
    if (e->cfg.scan_layout != 0) return fail(e, MSR_ERR_INVALID, "msr_enable_bf16: needs the row-major layout");
    if (e->emb_bf16) return MSR_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipError_t herr;
    const int64_t C = e->dense.n_chunks;
    // The image holds the rows NORMALISED and then rounded to bf16 (so a score needs no per-row scale and the error bound
    // of msr_batch.hip is about unit vectors), padded with 512 zero rows: the GEMM reads 256 rows from any tile start.
    const int64_t n_pad = C + 512;
    if ((herr = eng_malloc(e, &e->emb_bf16, (size_t)n_pad * MSR_DIM * 2)) != hipSuccess)
        return fail(e, MSR_ERR_NOMEM, "bf16 embeddings (%zu bytes): %s", (size_t)n_pad * MSR_DIM * 2, hipGetErrorString(herr));
    auto alloc = [&](void** p, size_t bytes) { return *p ? hipSuccess : eng_malloc(e, p, bytes); };
    const size_t QS = GM_SLICE;                             // the candidate scratch serves both the sweeps and the GEMM path
    if ((herr = alloc((void**)&e->bt_top_doc, (size_t)BT_SLICE * MSR_MAX_K * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bt_top_score, (size_t)BT_SLICE * MSR_MAX_K * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bt_top_n, (size_t)BT_SLICE * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bt_cand_doc, QS * MSR_SEL_CAP * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bt_cand_score, QS * MSR_SEL_CAP * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bt_cand_chunk, QS * MSR_SEL_CAP * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bt_cand_n, QS * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bf_ones, (size_t)C * 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bf_err, 4)) != hipSuccess ||
        (herr = alloc((void**)&e->bf_margin, QS * 4)) != hipSuccess)
        return fail(e, MSR_ERR_NOMEM, "bf16 path scratch: %s", hipGetErrorString(herr));
    HIP_TRY(e, hipMemsetAsync(e->bt_cand_n, 0, QS * 4, st));
    HIP_TRY(e, msr_unit_bf16_rows(e->dense.emb, e->dense.inv_norm, C, n_pad, e->emb_bf16, e->bf_err, st));
    HIP_TRY(e, msr_fill_f32(e->bf_ones, C, 1.0f, st));
    e->dense.emb_bf16 = e->emb_bf16;
    e->dense_bf16 = e->dense;
    e->dense_bf16.inv_norm = e->bf_ones;
    if (e->dense.wide_ok) {
        if ((herr = alloc(&e->bf_row_meta, (size_t)(C + 16) * 8)) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "bf16 row meta: %s", hipGetErrorString(herr));
        HIP_TRY(e, msr_pack_row_meta(e->chunk_doc, e->bf_ones, C, e->bf_row_meta, st));
        e->dense_bf16.row_meta = e->bf_row_meta;
    }
    // ---- GEMM path: needs the row tiles built at bind time (every document inside one 256-row tile) ----
    const bool ok = e->tiles_ok;
    const int n_tiles = e->n_tiles;
    const int grid = e->n_cus / 8 * 8;
    if (ok && n_tiles >= 64 && grid >= 32) {
        const int stride = (n_tiles + 31) / 32 * 32;
        if ((herr = alloc(&e->gm_qmat, (size_t)GM_SLICE * MSR_DIM * 2)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_qn, (size_t)GM_SLICE * MSR_DIM * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_tmax, (size_t)GM_SLICE * stride * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_tmax_t, (size_t)n_tiles * 2 * GM_SLICE * 4)) != hipSuccess ||     // [tile][query] (x 2: the diagnostic build's round-2 kernel stores two rows per tile)
            (herr = alloc((void**)&e->gm_thr, (size_t)GM_SLICE * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_thr2, (size_t)GM_SLICE * 4)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_flag, (size_t)GM_SLICE * 4)) != hipSuccess ||
            (herr = alloc(&e->gm_wgbuf, (size_t)grid * 8 * GM_WV_CAP * 16)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_wv_count, (size_t)grid * 8 * 4)) != hipSuccess ||
            (herr = alloc(&e->gm_pairs, (size_t)GM_SLICE * msr_gemm_pair_cap() * 8)) != hipSuccess ||
            (herr = alloc((void**)&e->gm_pair_n, (size_t)GM_SLICE * 4)) != hipSuccess)
            return fail(e, MSR_ERR_NOMEM, "GEMM path scratch: %s", hipGetErrorString(herr));
        HIP_TRY(e, hipMemsetAsync(e->gm_pair_n, 0, (size_t)GM_SLICE * 4, st));
        e->gemm = GemmIndex{e->emb_bf16, e->tile_row, n_tiles, e->n_cus, GM_SLICE, e->gm_qmat, e->gm_tmax, stride,
                            e->gm_tmax_t, e->gm_thr, e->gm_thr2, e->gm_flag, e->gm_wgbuf,
                            GM_WV_CAP, e->gm_wv_count, e->gm_pairs, e->gm_pair_n};
        e->gemm_ok = true;
    }
    return MSR_OK;
}

extern "C" int msr_tune(msr_engine* e, int32_t key, int32_t value) {
    if (!e) return MSR_ERR_INVALID;
#ifdef MSR_DIAG
    if (key == 100) { msr_gemm_set_dbg(value); return MSR_OK; }      // timing experiments of the diagnostic build
    if (key == 101) { msr_gemm_f32_set_dbg(value); return MSR_OK; }
    if (key == 102) { msr_bm25_set_dbg(value); return MSR_OK; }
#endif
    return fail(e, MSR_ERR_INVALID, "msr_tune: unknown key %d / value %d", key, value);
}

extern "C" int msr_batch_gemm_ok(const msr_engine* e) { return e && e->have_chunks && e->emb_bf16 && e->gemm_ok ? 1 : 0; }

extern "C" int msr_dense_topk_bf16(msr_engine* e, const float* q, int32_t n_queries, int32_t k,
                                   int32_t max_chunks_per_doc, int32_t* out_doc, float* out_score, int32_t* out_chunk,
                                   int32_t* out_n, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!e->have_chunks || !e->emb_bf16) return fail(e, MSR_ERR_NOT_BOUND, "msr_dense_topk_bf16: call msr_enable_bf16 first");
    if (n_queries < 0 || k < 1 || k > e->cfg.max_k || max_chunks_per_doc < 0 || !q || !out_doc || !out_score || !out_n)
        return fail(e, MSR_ERR_INVALID, "msr_dense_topk_bf16: bad argument (k=%d, max_k=%d)", k, e->cfg.max_k);
    if (n_queries == 0) return MSR_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int64_t N = e->dense.n_docs;
    // candidate margin: 2 eps_q from the measured rounding errors of the image and of each query (msr_batch.hip)
    // More than 128 queries: the tiled GEMM (msr_gemm.hip), GM_SLICE queries per pair of passes.  It needs whole
    // documents inside 256-row tiles, at least 2 k tiles (the sample bound) and no per-document row limit.
    if (e->gemm_ok && max_chunks_per_doc == 0 && n_queries > BT_SLICE && e->gemm.n_tiles >= 2 * k) {
        for (int q0 = 0; q0 < n_queries; q0 += GM_SLICE) {
            const int nq = std::min(GM_SLICE, n_queries - q0);
            HIP_TRY(e, msr_prep_queries(q + (int64_t)q0 * MSR_DIM, nq, e->gm_qn, nq, st));
            HIP_TRY(e, msr_batch_margin(e->gm_qn, nq, e->bf_err, e->bf_margin, st));
            hipEvent_t ev[4];
            const bool timed = e->timing && e->ev_count[2] < msr_engine::EV_RING && e->ev_count[3] < msr_engine::EV_RING;
            if (timed) {
                ev[0] = e->ev_start[3][e->ev_count[3]]; ev[1] = e->ev_stop[3][e->ev_count[3]];
                ev[2] = e->ev_start[2][e->ev_count[2]]; ev[3] = e->ev_stop[2][e->ev_count[2]];
            }
            // (candidates with the runs of their emitted rows: bt_cand_chunk carries first | len << 13 in, the arg-max row out)
            HIP_TRY(e, msr_gemm_candidates(e->gemm, e->dense, e->gm_qn, nq, k, e->bf_margin, e->bt_cand_doc, e->bt_cand_chunk,
                                           e->bt_cand_n, timed ? ev : nullptr, st));
            if (timed) { e->ev_count[2]++; e->ev_count[3]++; }
            HIP_TRY(e, msr_batch_rescore_rows(e->dense, e->gm_qn, nq, k, (const int32_t*)e->gemm.pairs, 2, msr_gemm_pair_cap(),
                                              e->bt_cand_doc, e->bt_cand_score, e->bt_cand_chunk, e->bt_cand_n,
                                              out_doc + (int64_t)q0 * k, out_score + (int64_t)q0 * k,
                                              out_chunk ? out_chunk + (int64_t)q0 * k : nullptr, out_n + q0, st));
        }
        return MSR_OK;
    }
    // 33..128 queries: K-split kernel (msr_dense_ks.hip).  Diagnostic build only (-DMSR_DIAG): MSR_BF16_WIDE=0 keeps
    // the wave-streaming kernel everywhere.
#ifdef MSR_DIAG
    static const bool wide_knob = [] { const char* v = getenv("MSR_BF16_WIDE"); return !v || atoi(v) != 0; }();
#else
    const bool wide_knob = true;
#endif
    const bool wide_able = wide_knob && e->dense.wide_ok && max_chunks_per_doc == 0;
    const int slice = wide_able && e->dense.wide_ok64 ? 128 : 64;
    for (int q0 = 0; q0 < n_queries; q0 += slice) {
        const int nq = std::min(slice, n_queries - q0);
        const bool wide = wide_able && nq > 32;
        // zero rows up to the query-block count of the kernel that runs
        const int nq_pad = wide ? (nq > 64 ? 128 : 64) : (nq + 15) / 16 * 16;
        HIP_TRY(e, msr_prep_queries(q + (int64_t)q0 * MSR_DIM, nq, e->qn, nq_pad, st));
        HIP_TRY(e, msr_batch_margin(e->qn, nq, e->bf_err, e->bf_margin, st));
        const bool timed = e->timing && e->ev_count[0] < msr_engine::EV_RING;
        if (timed) HIP_TRY(e, hipEventRecord(e->ev_start[0][e->ev_count[0]], st));
        // (the image holds unit rows: dense_bf16 carries inverse norms of 1)
        if (wide) HIP_TRY(e, msr_dense_scan_bf16_wide(e->dense_bf16, e->qn, nq, (float*)e->score_rows, st));
        else HIP_TRY(e, msr_dense_scan_bf16(e->dense_bf16, e->qn, nq, max_chunks_per_doc, (float*)e->score_rows, st));
        if (timed) {
            HIP_TRY(e, hipEventRecord(e->ev_stop[0][e->ev_count[0]], st));
            e->ev_count[0]++;
        }
        // k-th largest approximate score per query
        HIP_TRY(e, msr_select_topk(32, (const float*)e->score_rows, N, e->dense.score_stride, nq, k, e->sel, e->bt_top_doc,
                                   e->bt_top_score, e->bt_top_n, st));
        HIP_TRY(e, msr_batch_finish(e->dense, e->qn, nq, k, max_chunks_per_doc, e->bf_margin, (const float*)e->score_rows,
                                    e->bt_top_score, e->bt_top_n, e->bt_cand_doc, e->bt_cand_score, e->bt_cand_chunk,
                                    e->bt_cand_n, out_doc + (int64_t)q0 * k, out_score + (int64_t)q0 * k,
                                    out_chunk ? out_chunk + (int64_t)q0 * k : nullptr, out_n + q0, st));
    }
    return MSR_OK;
}

static int rerank_gather_impl(msr_engine* e, const char* fn, const float* q, int32_t n_queries, const int32_t* cand_doc,
                              const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base, int32_t max_chunks,
                              float* out_cos, int32_t* out_meta, int32_t q_per_block, int64_t block_stride, void* stream,
                              const RerankRecords* rec = nullptr) {
    if (!e) return MSR_ERR_INVALID;
    if (!e->have_chunks) return fail(e, MSR_ERR_NOT_BOUND, "%s: chunks not bound", fn);
    if (!q || !cand_doc || !cand_n || (!rec && (!out_cos || !out_meta))) return fail(e, MSR_ERR_INVALID, "%s: null argument", fn);
    int rc = rerank_args_ok(e, fn, n_queries, max_cand, max_chunks);
    if (rc) return rc;
    if (e->url_group && e->url_group_n != e->dense.n_docs)
        return fail(e, MSR_ERR_INVALID, "%s: doc meta bound for %lld docs, chunks for %lld", fn,
                    (long long)e->url_group_n, (long long)e->dense.n_docs);
    if (n_queries == 0) return MSR_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // ONE normalisation and ONE gather launch for up to max(max_queries, 128) queries (the scratch for the normalised queries)
    const int cap = std::max(e->cfg.max_queries, 128);
    for (int q0 = 0; q0 < n_queries; q0 += cap) {
        const int nq = std::min(cap, n_queries - q0);
        if (q0 % q_per_block != 0 && q_per_block < n_queries)
            return fail(e, MSR_ERR_INVALID, "%s: max_queries must be a multiple of queries_per_block for calls this large", fn);
        HIP_TRY(e, msr_prep_queries(q + (int64_t)q0 * MSR_DIM, nq, e->rr_qn, nq, st));
        const int64_t o = (int64_t)q0 * max_cand;
        const bool blocked = q_per_block < n_queries;
        float* co = blocked ? out_cos + (int64_t)(q0 / q_per_block) * block_stride : out_cos + o * MSR_RERANK_MAX_CHUNKS;
        int32_t* mo = blocked ? out_meta + (int64_t)(q0 / q_per_block) * block_stride : out_meta + o * 3;
        RerankRecords rr{nullptr, nullptr, nullptr, 0};
        if (rec) rr = RerankRecords{rec->out, rec->q_base + q0, rec->blk_off + (int64_t)q0 * ((max_cand + 7) / 8), rec->capacity};
        HIP_TRY(e, msr_rerank_gather(e->dense, e->url_group, e->rr_qn, nq, cand_doc + o, cand_n + q0, max_cand, doc_base,
                                     row_base, max_chunks, co, mo, blocked ? q_per_block : nq, blocked ? block_stride : 0, rr, st));
    }
    return MSR_OK;
}

extern "C" int msr_rerank_gather(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                                 const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base,
                                 int32_t max_chunks, float* out_cos, int32_t* out_meta, void* stream) {
    return rerank_gather_impl(e, "msr_rerank_gather", q, n_queries, cand_doc, cand_n, max_cand, doc_base, row_base, max_chunks,
                              out_cos, out_meta, n_queries > 0 ? n_queries : 1, 0, stream);
}

extern "C" int msr_rerank_gather_blocks(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                                        const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base,
                                        int32_t max_chunks, int32_t* out_blocks, int32_t queries_per_block,
                                        int64_t block_words, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (queries_per_block < 1 || block_words < (int64_t)queries_per_block * max_cand * (MSR_RERANK_MAX_CHUNKS + 3))
        return fail(e, MSR_ERR_INVALID, "msr_rerank_gather_blocks: a block of %lld words cannot hold %d queries", (long long)block_words,
                    queries_per_block);
    // block b = [cos of its queries: qpb x max_cand x 10 | meta: qpb x max_cand x 3 | padding]
    return rerank_gather_impl(e, "msr_rerank_gather_blocks", q, n_queries, cand_doc, cand_n, max_cand, doc_base, row_base,
                              max_chunks, (float*)out_blocks, out_blocks + (int64_t)queries_per_block * max_cand * MSR_RERANK_MAX_CHUNKS,
                              queries_per_block, block_words, stream);
}

extern "C" int msr_rerank_plan(msr_engine* e, int32_t n_queries, const int32_t* cand_doc, const int32_t* cand_n,
                               int32_t max_cand, const int32_t* shard_bounds, int32_t n_shards, int32_t my_shard,
                               int32_t queries_per_shard, int32_t* counts, int32_t* send_base, int32_t* send_blk,
                               int32_t* recv_off, int32_t* pair, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!cand_doc || !cand_n || !shard_bounds || !counts || !send_base || !send_blk || !recv_off || !pair || n_queries < 0 ||
        max_cand < 1 || max_cand > 1024 || n_shards < 1 || n_shards > 64 || my_shard < 0 || my_shard >= n_shards ||
        queries_per_shard < 1 || (int64_t)queries_per_shard * n_shards < n_queries)
        return fail(e, MSR_ERR_INVALID, "msr_rerank_plan: bad argument (n_shards=%d, my_shard=%d, queries_per_shard=%d)", n_shards,
                    my_shard, queries_per_shard);
    if (n_queries == 0) return MSR_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_rerank_plan_run(n_queries, cand_doc, cand_n, max_cand, shard_bounds, n_shards, my_shard, queries_per_shard,
                                   counts, send_base, send_blk, recv_off, pair, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_rerank_gather_records(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                                         const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base,
                                         int32_t max_chunks, const int32_t* send_base, const int32_t* send_blk,
                                         int32_t* out_records, int64_t capacity_records, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!send_base || !send_blk || !out_records || capacity_records < 0)
        return fail(e, MSR_ERR_INVALID, "msr_rerank_gather_records: bad argument");
    const RerankRecords rec{out_records, send_base, send_blk, capacity_records};
    return rerank_gather_impl(e, "msr_rerank_gather_records", q, n_queries, cand_doc, cand_n, max_cand, doc_base, row_base,
                              max_chunks, nullptr, nullptr, n_queries > 0 ? n_queries : 1, 0, stream, &rec);
}

extern "C" int msr_rerank_scatter(msr_engine* e, const int32_t* records, int64_t capacity_records, const int32_t* counts,
                                  const int32_t* recv_off, int32_t n_shards, int32_t n_queries, int32_t queries_per_shard,
                                  int32_t first_query, int32_t n_my_queries, int32_t max_cand, float* out_cos, int32_t* out_meta,
                                  void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!records || capacity_records < 0 || !counts || !recv_off || !out_cos || !out_meta || n_shards < 1 || n_shards > 64 || max_cand < 1 ||
        max_cand > 1024 || n_my_queries < 0 || n_my_queries > queries_per_shard || first_query < 0 ||
        first_query + n_my_queries > n_queries)
        return fail(e, MSR_ERR_INVALID, "msr_rerank_scatter: bad argument (n_shards=%d, first_query=%d, n_my_queries=%d)", n_shards,
                    first_query, n_my_queries);
    if (n_my_queries == 0) return MSR_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_rerank_scatter_run(records, capacity_records, counts, recv_off, n_shards, n_queries, queries_per_shard, first_query,
                                      n_my_queries, max_cand, out_cos, out_meta, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_rerank_fuse(msr_engine* e, int32_t n_queries, const int32_t* cand_doc, const double* cand_bm25,
                               const int32_t* cand_n, int32_t max_cand, const float* cos, const int32_t* meta,
                               const msr_rerank_params* params, int32_t* out_doc, double* out_score,
                               double* out_orig, int32_t* out_chunk, int32_t* out_n, int32_t* out_rows, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!cand_doc || !cand_bm25 || !cand_n || !cos || !meta || !params || !out_doc || !out_score || !out_orig ||
        !out_chunk || !out_n || !out_rows)
        return fail(e, MSR_ERR_INVALID, "msr_rerank_fuse: null argument");
    int rc = rerank_args_ok(e, "msr_rerank_fuse", n_queries, max_cand, params->max_chunks);
    if (rc) return rc;
    if (n_queries == 0) return MSR_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const RerankParams p{params->smoothing, params->max_boost, params->max_decay, params->max_chunks};
    HIP_TRY(e, msr_rerank_fuse_run(n_queries, cand_doc, cand_bm25, cand_n, max_cand, p, cos, meta, out_doc, out_score,
                                   out_orig, out_chunk, out_n, out_rows, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_rerank_combine(msr_engine* e, const float* cos_parts, const int32_t* meta_parts, int32_t n_parts,
                                  int64_t part_stride_bytes, int32_t n_queries, int32_t max_cand, float* out_cos,
                                  int32_t* out_meta, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!cos_parts || !meta_parts || !out_cos || !out_meta || n_parts < 1 || n_parts > 64 || n_queries < 0 || max_cand < 1 ||
        max_cand > 1024 || part_stride_bytes < 0 || (part_stride_bytes & 3) || (n_parts > 1 && part_stride_bytes == 0))
        return fail(e, MSR_ERR_INVALID, "msr_rerank_combine: bad argument (n_parts=%d, max_cand=%d)", n_parts, max_cand);
    if (n_queries == 0) return MSR_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int64_t rows = (int64_t)n_queries * max_cand;
    HIP_TRY(e, msr_or_parts(cos_parts, n_parts, part_stride_bytes, rows * MSR_RERANK_MAX_CHUNKS, out_cos, (hipStream_t)stream));
    HIP_TRY(e, msr_or_parts(meta_parts, n_parts, part_stride_bytes, rows * 3, out_meta, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_rerank(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                          const double* cand_bm25, const int32_t* cand_n, int32_t max_cand,
                          const msr_rerank_params* params, int32_t* out_doc, double* out_score, double* out_orig,
                          int32_t* out_chunk, int32_t* out_n, int32_t* out_rows, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!params) return fail(e, MSR_ERR_INVALID, "msr_rerank: null params");
    // unsharded convenience: gather (this engine owns every document) + fuse, slice by slice
    const int slice = std::max(e->cfg.max_queries, 128);    // the gather scratch holds this many queries
    for (int q0 = 0; q0 < n_queries; q0 += slice) {
        const int nq = std::min(slice, n_queries - q0);
        const int64_t o = (int64_t)q0 * max_cand;
        int rc = msr_rerank_gather(e, q ? q + (int64_t)q0 * MSR_DIM : nullptr, nq, cand_doc ? cand_doc + o : nullptr,
                                   cand_n ? cand_n + q0 : nullptr, max_cand, 0, 0, params->max_chunks, e->rerank_cos,
                                   e->rerank_meta, stream);
        if (rc) return rc;
        rc = msr_rerank_fuse(e, nq, cand_doc + o, cand_bm25 ? cand_bm25 + o : nullptr, cand_n + q0, max_cand,
                             e->rerank_cos, e->rerank_meta, params, out_doc ? out_doc + o : nullptr,
                             out_score ? out_score + o : nullptr, out_orig ? out_orig + o : nullptr,
                             out_chunk ? out_chunk + o : nullptr, out_n ? out_n + q0 : nullptr,
                             out_rows ? out_rows + q0 : nullptr, stream);
        if (rc) return rc;
    }
    return MSR_OK;
}

extern "C" int msr_merge_topk_payload(msr_engine* e, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                                      const int32_t* in_payload, int32_t n_parts, int64_t part_stride_bytes,
                                      int32_t n_queries, int32_t k, int32_t score_bits, int32_t* out_doc, void* out_score,
                                      int32_t* out_n, int32_t* out_payload, void* stream) {
    if (!e) return MSR_ERR_INVALID;
    if (!in_doc || !in_score || !in_n || !out_doc || !out_score || !out_n || n_parts < 1 || n_parts > 64 ||
        n_queries < 0 || k < 1 || k > MSR_MAX_K || (score_bits != 32 && score_bits != 64) || (int64_t)n_parts * k > 8192 ||
        (in_payload != nullptr) != (out_payload != nullptr) || part_stride_bytes < 0 || (part_stride_bytes & 7))
        return fail(e, MSR_ERR_INVALID, "msr_merge_topk: bad argument (n_parts=%d, k=%d)", n_parts, k);
    if (n_queries == 0) return MSR_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, msr_merge_lists(score_bits, in_doc, in_score, in_n, in_payload, n_parts, part_stride_bytes, n_queries, k, out_doc,
                               out_score, out_n, out_payload, (hipStream_t)stream));
    return MSR_OK;
}

extern "C" int msr_merge_topk(msr_engine* e, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                              int32_t n_parts, int32_t n_queries, int32_t k, int32_t score_bits, int32_t* out_doc,
                              void* out_score, int32_t* out_n, void* stream) {
    return msr_merge_topk_payload(e, in_doc, in_score, in_n, nullptr, n_parts, 0, n_queries, k, score_bits, out_doc, out_score,
                                  out_n, nullptr, stream);
}
