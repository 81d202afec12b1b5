// Internal (C++) interfaces between the translation units of libmsretr.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSR_DIM 768
#define MSR_SEL_BINS 4096      // 12-bit radix digits
#define MSR_SEL_CAP 4096       // candidate capacity per query of the exact final sort

// ---- K4: radix-select top-k over a dense score row per query -------------------------------------
struct SelState {
    uint64_t pref_hi, mask_hi;   // resolved digits of the score part of the key
    uint32_t pref_lo, mask_lo;   // resolved digits of the (~doc index) part of the key
    int32_t k_rem;               // how many more are needed from inside the current tie bin
    int32_t n_above;             // elements strictly above the resolved prefix
    int32_t done;
    int32_t n_sel;               // number of entries that will be emitted (min(k, #valid))
};

struct SelScratch {
    uint32_t* hist;      // [nq][MSR_SEL_BINS], all zero between calls
    SelState* state;     // [nq]
    uint64_t* cand_hi;   // [nq][MSR_SEL_CAP]
    uint32_t* cand_lo;   // [nq][MSR_SEL_CAP]
    int32_t* cand_n;     // [nq], all zero between calls
    const int32_t* gate; // optional device word(s): when non-null and the word is 0 every kernel of the select returns at once
                         // (fallback launches that are only needed when an earlier kernel raised the flag)
    int gate_per64;      // 0: gate[0] decides for all queries; 1: gate[q / 64] decides for query q (one select over several
                         // 64-query slices of which only some need the fallback)
};

// Select the top-k of scores[q][0..n) (row stride `stride` elements) for q in [0, nq).
// score_bits 32: float scores / float out_score; 64: double.  Rows of out_* have stride k.
// Order: score desc, index asc.  Entries past out_n[q] are (-1, -inf).
hipError_t msr_select_topk(int score_bits, const void* scores, int64_t n, int64_t stride, int nq, int k,
                           const SelScratch& sc, int32_t* out_doc, void* out_score, int32_t* out_n,
                           hipStream_t stream);

// The same over segmented LISTS: row q (stride elements apart) is cut into n_seg segments seg_stride elements apart; segment s
// holds counts[q * n_seg + s] pairs (scores, idx) in any order, from its first position on.
// win_base (nullable, [nq]): the first histogram pass takes a WINDOW of the 20-bit key prefixes (sign, exponent, 8 mantissa
// bits) instead of the first 12-bit digit -- bin = clamp(prefix - win_base[q], 0, 4095) -- and resolves 20 bits at once: one
// streaming pass instead of two when an upper bound of the scores is known (msr_bm25_window); a k-th key in bin 0 or 4095
// (the bound was wrong, or the k-th score is 2^-15 of it) sends the query down the general in-kernel path: exact either way.
hipError_t msr_select_topk_list(const double* scores, const int32_t* idx, const int32_t* counts, int n_seg, int64_t seg_stride,
                                int64_t stride, int nq, int k, const SelScratch& sc, int32_t* out_doc,
                                double* out_score, int32_t* out_n, hipStream_t stream, const uint64_t* win_base = nullptr);

// Merge lists: in_* [n_parts][nq][k]; see msretr.h msr_merge_topk.  in_pay / out_pay (nullable): a 32-bit payload per entry
// that travels with it (the arg-max chunk row of a dense result).  part_stride_bytes != 0: part p of EVERY input array starts
// that many bytes after part p - 1 (the receive buffer of one all-gather: [rank][packed segments]), else contiguous.
hipError_t msr_merge_lists(int score_bits, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                           const int32_t* in_pay, int n_parts, int64_t part_stride_bytes, int nq, int k, int32_t* out_doc,
                           void* out_score, int32_t* out_n, int32_t* out_pay, hipStream_t stream);

// error text of handle-less entry points, read back with msr_last_error(NULL) (msr_engine.hip)
int msr_fail_global(int code, const char* fmt, ...);

// ---- K1: BM25 term-at-a-time ----------------------------------------------------------------------
// One posting as the scoring kernel streams it (engine-owned copy, built at bind): the document and the posting's
// tf_component = (tf (k1 + 1)) / (tf + k1 (1 - b + b dl / avgdl)) (indexer/bm25_indexer.py:473-475), which depends on
// (tf, document) only -- evaluated ONCE with the reference's own operations, so the kernel neither divides nor looks the
// document length up.
struct Bm25Post {              // 12 bytes, 4-byte aligned: one global_load_dwordx3 per posting
    int32_t doc;
    uint32_t comp_lo, comp_hi; // the float64 tf_component
};
struct Bm25Index {
    const int64_t* term_off;
    const int32_t* post_doc;
    const int32_t* post_tf;
    const int32_t* doc_len;
    const float* idf;
    int64_t n_terms, n_postings, n_docs;
    double avgdl, k1, b;
    // skip table (engine-owned, built at bind): for the terms with long posting lists, where each document
    // tile starts inside the list, so a wave finds its slice with two loads instead of a search
    const int32_t* heavy_id;   // [n_terms]: row of tile_off, or -1
    const uint32_t* tile_off;  // [n_heavy][n_tiles + 1], relative to term_off[t]
    int32_t n_tiles;
    const Bm25Post* post;      // [n_postings + 1] {doc, tf_component}: what the scoring kernel streams; the last entry is the
                               // sentinel {-1, 0.0} that lanes without a posting load
    // dense tf_component tables of the long lists with NEGATIVE idf (document frequency above half the corpus): such a
    // term can only lower a score, so with min_score >= 0 a document it alone matches is never a candidate; its list is not
    // streamed at all, its contribution is looked up for the documents the other terms touch (msr_bm25.hip)
    const int32_t* dense_id;   // [n_terms]: row of dense_comp, or -1 (null: no table)
    const double* dense_comp;  // [n_dense][dense_stride]: tf_component of (term, document), 0.0 = the document lacks the term;
                               // dense_stride > the padded document count: the last entry of a row is always 0.0
    int64_t dense_stride;
};
hipError_t msr_bm25_dnorm(const int32_t* doc_len, int64_t n_docs, int64_t n_pad, double k1, double b, double avgdl, double* out,
                          hipStream_t stream);
// out[i] = {post_doc[i], tf_component(post_tf[i], dnorm[post_doc[i]])} for i < n, out[n] = {-1, 0.0}
hipError_t msr_bm25_post_comp(const int32_t* post_doc, const int32_t* post_tf, const double* dnorm, double k1, int64_t n,
                              Bm25Post* out, hipStream_t stream);
// dense_comp row h <- the tf_components of term dense_terms[h] scattered by document (rows zeroed by the caller)
hipError_t msr_bm25_build_dense(const Bm25Index& ix, const int32_t* dense_terms, int n_dense, double* dense_comp,
                                int64_t dense_stride, hipStream_t stream);
constexpr int MSR_BM25_TILE = 1024;          // documents per BM25 tile
constexpr int MSR_BM25_HEAVY_DF = 2048;      // posting lists at least this long get a skip-table row
constexpr int MSR_BM25_MAX_DENSE = 64;       // at most this many dense tf_component tables
hipError_t msr_bm25_build_skip(const Bm25Index& ix, const int32_t* heavy_terms, int n_heavy, uint32_t* tile_off,
                               hipStream_t stream);
// Candidate lists: exactly the documents touched by a posting whose score is >= min_score, as pairs (cand_score, cand_doc)
// in SEGMENTS: row q (stride n_docs) is cut into *n_seg segments of *seg_stride documents (one per work item of the kernel:
// a span of tiles); segment s holds seg_n[q * n_seg + s] pairs from position s * seg_stride on, in no particular order.
// Every count is written by the kernel (no initialisation, no atomics).  seg_n: msr_bm25_max_segments(n_docs) words per query.
int msr_bm25_max_segments(int64_t n_docs);
hipError_t msr_bm25_window(const Bm25Index& ix, const int32_t* q_term_off, const int32_t* q_terms, const int32_t* q_qtf,
                           int q_first, int nq, uint64_t* out /*[nq]: see msr_select_topk_list*/, hipStream_t stream);
hipError_t msr_bm25_scores(const Bm25Index& ix, const int32_t* q_term_off, const int32_t* q_terms,
                           const int32_t* q_qtf, int q_first, int nq, double min_score, double* cand_score,
                           int32_t* cand_doc, int32_t* seg_n, int* n_seg, int64_t* seg_stride, hipStream_t stream);

// *flag (device) <- 0x7F7F7F7F if the CSR is well formed, else the lowest violated rule number (msr_bm25.hip).
hipError_t msr_bm25_validate(const Bm25Index& ix, int32_t* flag, hipStream_t stream);

// ---- K2/K3: dense scan + per-document max ---------------------------------------------------------
constexpr int MSR_WIDE_RING = 128;           // documents in the K-split kernels' LDS ring of maxima
struct DenseIndex {
    const float* emb;          // row-major [n_chunks][768] or interleaved image
    const int32_t* doc_off;    // [n_docs+1]
    const int32_t* chunk_doc;  // [n_chunks]
    const float* inv_norm;     // [n_chunks]
    const int32_t* span_doc;   // [n_spans+1] document boundaries of the workgroup spans (scan variant 1)
    int64_t n_chunks, n_docs;
    int64_t score_stride;      // elements between the score rows of consecutive queries (n_docs rounded up to 32:
                               // every row starts on a 128 B line)
    int32_t n_spans;
    int32_t layout;            // 0 row-major, 1 interleaved
    const int32_t* wspan_doc;  // [n_wspans+1] document boundaries of the per-wave spans (scan variants 2, 3)
    int32_t n_wspans;
    const int32_t* wspan12_doc; // the same for 12 waves per CU (scan variants 5, 6)
    int32_t n_wspans12;
    void* qimg;                // engine scratch: query image in MFMA-fragment order (<= 256 KB)
    const void* emb_bf16;      // bf16 [n_chunks][768] copy of emb for the batched path, or null
    const void* emb_presplit;  // scan_variant 15 only: the rows as f16 hi/lo pieces (msr_presplit_rows), same size as emb
    const void* row_meta;      // [n_chunks + 16] x {int32 document, float inverse norm} (last row repeated as padding: a row group may stick out by 15 rows),
                               // one load per unit for the K-split kernels; null when !wide_ok
    int32_t wide_ok;           // any 32 consecutive rows (on 16-row group boundaries) span <= MSR_WIDE_RING - 32 documents
                               // (the K-split kernels' ring of per-document maxima cannot wrap onto live slots) and any
                               // 16-row group <= 32 documents (at most two finished blocks per unit)
    int32_t wide_ok64;         // the same with a ring of 64 documents (128-query bf16 sweeps)
    const int32_t* gate;       // optional device word: when non-null and *gate == 0 the scan / best-chunk kernels return at once
    int32_t variant;           // scan kernel: 14 = K-split kernel, f16-split products, <= 64 queries per sweep (default when
                               // the row norms and wide_ok allow it; 13 = the same for 17..64 queries only), 7 = wave
                               // streaming with f16-split products, 2 = wave streaming, exact f32 MFMA (default otherwise),
                               // 1 = super-tile kernel of the first profile, others: A/B variants (msr_dense.hip)
    int32_t gate_per64;        // best-chunk kernel only: 1 = gate[q / 64] decides for query q (0: gate[0] for all)
};
// qn: [ceil16(nq)][768] normalised queries (zero rows as padding).
// docscore[q][ix.score_stride] <- max cosine over the document's chunks (-inf for chunk-less documents).
hipError_t msr_dense_scan(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                          hipStream_t stream);
// bf16 candidate scan (<= 64 queries per sweep); qn as above with ceil16(nq) rows.
hipError_t msr_dense_scan_bf16(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                               hipStream_t stream);
// Wide sweeps (msr_dense_ks.hip): f16-split products over f32 rows for <= 64 queries per pass; bf16 rows for <= 64,
// or <= 128 with ix.wide_ok64.
// Both need ix.wide_ok and have no per-document row limit (max_chunks = 0).
hipError_t msr_dense_scan_wide(const DenseIndex& ix, const float* qn, int nq, float* docscore, hipStream_t stream);
// gated fallback of the streaming pass: all 64-query slices of a call in one launch (msr_dense_ks.hip)
size_t msr_ksplit_slice_image_bytes();
hipError_t msr_dense_scan_slices(const DenseIndex& ix, const float* qn, int nq, float* docscore, void* qimg_slices,
                                 hipStream_t stream);
hipError_t msr_dense_scan_wide_exact(const DenseIndex& ix, const float* qn, int nq, float* docscore, hipStream_t stream);
hipError_t msr_dense_scan_bf16_wide(const DenseIndex& ix, const float* qn, int nq, float* docscore,
                                    hipStream_t stream);
// Query image in MFMA-fragment order for n_blocks x 16 queries (mode: 0 f32, 1 bf16, 2 f16 hi/lo pieces).
hipError_t msr_build_qimage(int mode, const float* qn, int n_blocks, void* qimg, hipStream_t stream);
// dst (n_rows x 768 x 4 bytes) <- the rows as f16 hi/lo pieces in the slots the f16-split scan loads them from
hipError_t msr_presplit_rows(const float* src, int64_t n_rows, void* dst, hipStream_t stream);
// row_meta[i] = {chunk_doc[min(i, n-1)], inv_norm[min(i, n-1)]} for i in [0, n + 16)
hipError_t msr_pack_row_meta(const int32_t* chunk_doc, const float* inv_norm, int64_t n, void* row_meta, hipStream_t stream);
hipError_t msr_fill_f32(float* dst, int64_t n, float value, hipStream_t stream);
// out2 (device, 2 words) <- bit patterns of min and max of inv_norm[0..n)
hipError_t msr_inv_norm_range(const float* inv_norm, int64_t n, uint32_t* out2, hipStream_t stream);
hipError_t msr_prep_queries(const float* q, int nq, float* qn, int nq_pad, hipStream_t stream);
hipError_t msr_fill_chunk_doc(const int32_t* doc_off, int64_t n_docs, int32_t* chunk_doc, hipStream_t stream);
hipError_t msr_row_inv_norm(const float* emb, int64_t n_rows, float* inv_norm, hipStream_t stream);
hipError_t msr_interleave(const float* src, int64_t n_rows, float* dst, hipStream_t stream);
// For each (q, r < out_n[q]): out_chunk[q][r] = row of the first maximum cosine among the winner's chunks.
hipError_t msr_best_chunk(const DenseIndex& ix, const float* qn, int nq, int k, int max_chunks,
                          const int32_t* out_doc, const int32_t* out_n, int32_t* out_chunk, hipStream_t stream);

// ---- K6: rerank / fuse ----------------------------------------------------------------------------
struct RerankParams {
    double smoothing, max_boost, max_decay;
    int32_t max_chunks;
};
// The compact form of (A)'s output (sharded runs): a record of MSR_RERANK_RECORD_WORDS words per OWNED slot, at record number
// q_base[q] + blk_off[q][slot / 8] + (owned slots before it in its block of 8).  out == nullptr: the dense arrays.
#define MSR_RERANK_RECORD_WORDS 16
struct RerankRecords {
    int32_t* out;
    const int32_t* q_base;
    const int32_t* blk_off;
    int64_t capacity;            // records `out` holds: a record at or past it is dropped (a caller's sizing error must not write outside)
};
// (A) cosines + (rows, url group, first row) of the candidates this shard owns; zeros for the others.
hipError_t msr_rerank_gather(const DenseIndex& ix, const int32_t* url_group, const float* qn, int nq,
                             const int32_t* cand_doc, const int32_t* cand_n, int max_cand, int doc_base,
                             int row_base, int max_chunks, float* cos_out /*[nq][max_cand][10]*/,
                             int32_t* meta /*[nq][max_cand][3]*/, int q_per_block, int64_t block_stride,
                             const RerankRecords& rec, hipStream_t stream);
// counts[s][q]: candidates of query q that shard s owns (bounds[s] <= doc < bounds[s + 1]); send_base / blk_off: see
// RerankRecords (for shard `my`); recv_off[s][j]: first record of (source s, query my * qps + j) in the receive buffer of an
// all-to-all whose (source, destination) blocks hold pair[s][o] records
hipError_t msr_rerank_plan_run(int nq, const int32_t* cand_doc, const int32_t* cand_n, int max_cand, const int32_t* bounds,
                               int n_shards, int my, int qps, int32_t* counts, int32_t* send_base, int32_t* blk_off,
                               int32_t* recv_off, int32_t* pair, hipStream_t stream);
hipError_t msr_rerank_scatter_run(const int32_t* records, int64_t capacity, const int32_t* counts, const int32_t* recv_off, int n_shards, int nq,
                                  int qps, int q_first, int n_mine, int max_cand, float* cos_out, int32_t* meta_out,
                                  hipStream_t stream);
// (q_per_block / block_stride: query q's rows start (q / q_per_block) * block_stride 32-bit words + (q % q_per_block) rows
// into cos_out / meta -- the blocks of an all-to-all send buffer; one contiguous array: q_per_block >= nq, any stride)
// (B) the float64 chain; needs no index.
hipError_t msr_rerank_fuse_run(int nq, const int32_t* cand_doc, const double* cand_bm25, const int32_t* cand_n,
                               int max_cand, const RerankParams& p, const float* cos_in, const int32_t* meta,
                               int32_t* out_doc, double* out_score, double* out_orig, int32_t* out_chunk,
                               int32_t* out_n, int32_t* out_rows, hipStream_t stream);

// domain diversification of fused lists (reranker_api.py:178-236): f_* as msr_rerank_fuse_run leaves them; doc_domain[d] =
// domain id of document d, -1 = rejected from responses (nullptr: every document accepted, its own domain)
hipError_t msr_diversify_run(int nq, const int32_t* f_doc, const double* f_score, const double* f_orig, const int32_t* f_chunk,
                             const int32_t* f_n, int max_cand, const int32_t* doc_domain, int64_t n_domain_docs, int top_k,
                             double threshold, int diversify, int32_t* out_doc, double* out_score, double* out_orig,
                             int32_t* out_chunk, int32_t* out_n, hipStream_t stream);

// out[0 .. n_words) = OR over the n_parts arrays in + p * part_stride_bytes (32-bit words)
hipError_t msr_or_parts(const void* in, int n_parts, int64_t part_stride_bytes, int64_t n_words, void* out, hipStream_t stream);

// ---- K5 as a tiled GEMM (msr_gemm.hip): candidates for hundreds to thousands of queries per call ------------------
struct GemmIndex {
    const void* emb_n;         // bf16 [n_chunks + pad][768]: the rows NORMALISED, then rounded (zero rows as padding)
    const int32_t* tile_row;   // [n_tiles + 1]: row tiles of <= 256 rows cut at document boundaries
    int32_t n_tiles;
    int32_t n_cus;
    int32_t max_queries;       // queries per call the scratch below is sized for (multiple of 256)
    void* qmat;                // bf16 [max_queries][768]
    float* tmax;               // [max_queries][tmax_stride] tile maxima, one row per query (input of the top-k select)
    int32_t tmax_stride;
    float* tmax_t;             // [n_tiles][max_queries] as the pass stores them (one row of maxima per tile)
    float* thr; float* thr2;   // [max_queries] emission threshold (sample bound) / final threshold (all tiles)
    int32_t* flag;             // [max_queries] 1: the sample could not bound this query (rerun on the exact path)
    void* wgbuf;               // [n_workgroups * 8 waves][wv_cap] x 16 B emitted (row, query, score, tile)
    int32_t wv_cap;
    int32_t* wv_count;         // [n_workgroups * 8]
    void* pairs;               // [max_queries][msr_gemm_pair_cap()] x 8 B (row, score) after the final threshold
    int32_t* pair_n;           // [max_queries], zero between calls
};
int msr_gemm_pair_cap();
// pieces of the candidate pipeline shared with the f32-class GEMM (msr_gemm_f32.hip)
hipError_t msr_gemm_tmax(const float* tmax_t, int n_j, int parts, int nq_pad, float* out, int out_stride, hipStream_t stream);
// thr[q] = k-th largest valid value of row q of tmax ([nq][stride], n values per row) - margin[q]; one launch
hipError_t msr_gemm_kth(const float* tmax, int n, int stride, int nq, int nq_pad, int k, const float* margin, float* thr,
                        int32_t* flag, hipStream_t stream, float short_val = __builtin_inff(), float margin_scale = 1.0f);
// (short_val: what thr[q] becomes when row q holds fewer than k valid values -- and for the padding rows q >= nq;
// margin_scale: thr = k-th value - margin_scale * margin[q])
hipError_t msr_gemm_bucket(const void* wvbuf, int wv_cap, const int32_t* wv_count, int n_waves, const float* thr2,
                           void* pairs, int32_t* pair_n, hipStream_t stream);
void msr_bm25_set_dbg(int v);       // honoured by -DMSR_DIAG builds only
void msr_gemm_set_dbg(int v);       // honoured by -DMSR_DIAG builds only
// dst[r] = bf16(src[r] * inv_norm[r]) (inv_norm null: 1), rows n_rows .. n_pad - 1 zero
// err_max (device word, nullable) <- bits of max_r || bf16(u_r) - u_r ||, u_r the normalised row
hipError_t msr_unit_bf16_rows(const float* src, const float* inv_norm, int64_t n_rows, int64_t n_pad, void* dst,
                              uint32_t* err_max, hipStream_t stream);
// margin[q] = 2 (dE (1 + dq) + dq) + slack from the MEASURED rounding errors of the image (err_max) and of query q
hipError_t msr_batch_margin(const float* qn, int nq, const uint32_t* err_max, float* margin, hipStream_t stream);
// qn: [nq][768] normalised f32 queries, nq <= g.max_queries.  Fills cand_doc[q][MSR_SEL_CAP] / cand_n[q] (cand_n =
// MSR_SEL_CAP + 1: overflow, rerun that query on the exact path) and cand_first_len[q][MSR_SEL_CAP] (the run of the
// candidate's emitted rows in g.pairs: first | len << 13, msr_gemm_pair_cap() entries of 2 words per query) for
// msr_batch_rescore_rows.  ev (nullable): 4 events recorded around the sample pass (0, 1) and the emit pass (2, 3).
hipError_t msr_gemm_candidates(const GemmIndex& g, const DenseIndex& ix, const float* qn, int nq, int k, const float* margin,
                               int32_t* cand_doc, int32_t* cand_first_len, int32_t* cand_n, hipEvent_t* ev,
                               hipStream_t stream);

// Rows a workgroup of the 256-query streaming kernel loads per tile visit, from the tile's first row on and whatever the
// tile's own length: 8 waves x 32 rows.  Sizes the zero padding behind the last tile of the fragment-order copy (msr_engine.hip).
constexpr int MSR_STREAM256_TILE_ROWS = 256;

// Arguments of the streaming passes (msr_gemm_f32.hip): one persistent workgroup per CU walks row tiles (<= 256 rows, cut at
// document boundaries), rows through a register ring, the query image through LDS.
struct StreamArgs {
    const char* E;             // rows: f32 [n_rows][768] (caller's matrix: NOT padded, the last tile clamps its row index), or the
                               // bf16 unit-row image (msr_stream256_bf16_launch)
    const float* inv_pad;      // [n_rows + 512] inverse norms (engine-owned padded copy; unused for the bf16 image)
    const char* qimg;          // query image: [group][24 K steps][128 or 256 queries][64 B], chunk-swizzled
    const int32_t* tile_row;   // [n_tiles + 1]
    const int32_t* tile_trow;  // [n_tiles] first row of each tile in the fragment-order copy (a multiple of 16; TILED kernels only)
    int64_t n_rows;
    int t_first, t_stride, t_count;
    float* tmax_t;             // f32 rows: [t_count][8 waves][queries of a launch]; bf16 rows: [t_count][256 nt] (joined in LDS)
    const float* thr;          // emit thresholds of the pass' queries (+inf: never)                    -- emit pass only
    void* wvbuf; int wv_cap; int32_t* wv_count;   // per-wave emission buffers {row, query, score bits, tile} (int4)   -- emit pass only
    int q_base;                // number of the pass' first query within the call (the query field of an entry is global)
    int append;                // emit pass: continue behind the entries earlier groups left in the wave buffers
    int nt;                    // 256-query kernel: query groups that share a tile sequence in ONE launch (1: none)
    int dbg;                   // -DMSR_DIAG builds only (timing experiments, wrong results)
};
// the 256-query kernel over the bf16 unit-row image, nt groups of 256 queries per launch (grid: a multiple of 8, >= 8 nt)
hipError_t msr_stream256_bf16_launch(bool emit, const StreamArgs& a, int grid, hipStream_t stream);
// its query image: bf16, [n_groups][24][256][64 B]
hipError_t msr_stream256_bf16_qimage(const float* qn, int nq, int n_groups, void* qimg, hipStream_t stream);

// ---- K2 for 65 .. 128 queries: one streaming pass over the f32 rows, f16 filter + exact f32 finish (msr_gemm_f32.hip) ----
struct GemmF32Index {
    const int32_t* tile_row;   // [n_tiles + 1] (the tile table of the bf16 GEMM: whole documents, <= 256 rows)
    int32_t n_tiles;
    int32_t n_cus;
    int32_t max_groups;        // groups of 128 queries one call may hold (the per-query arrays below are sized 128 x this)
    const float* inv_pad;      // [n_chunks + 512] inverse row norms, padded with 1
    void* qimg;                // [max_groups] x 24 x 8 KB query images (f16)
    float* tmax_t;             // [n_tiles][8 waves][128], or [256 max_nt] when max_groups >= 2 (the 256-query kernel)
    float* tmax;               // [128 max_groups][tmax_stride]
    int32_t tmax_stride;
    float* thr; float* thr2; int32_t* flag;                 // [128]
    void* wvbuf; int32_t wv_cap; int32_t* wv_count;          // [n_cus * 8][wv_cap] x 16 B / [n_cus * 8]
    void* pairs; int32_t* pair_n;                            // [128][4096] x 8 B / [128], zero between calls
    uint32_t* err_max; float* margin;                        // measured f16 rounding error of the rows (1 word) / margin [128]
    int32_t* cand_doc; float* cand_score; int32_t* cand_chunk; int32_t* cand_n;   // [128][MSR_SEL_CAP] x 3 / [128] (zero between calls)
    int32_t max_nt;            // groups of 256 queries one launch of the 256-query kernel may serve (tmax_t holds 256 x this per row)
    const void* emb_tiled;     // fragment-order copy of the f32 rows for the 256-query kernel (nullptr: none), see msr_tile_rows
    const int32_t* tile_trow;  // [n_tiles] first row of each tile in that copy
    const void* emb_f16;       // row-major f16 image of the rows (as the pass converts them; [n_chunks + 512][768], zero padded)
                               // for launches of several query groups (nullptr: none), see msr_f16_rows
};
// emb_tiled <- fragment-order copy of emb: tile t's rows at tile_trow[t] (multiples of 16), 16-row groups x 24 K steps x 2 KB
hipError_t msr_f16_rows(const float* emb, int64_t n_rows, int64_t n_pad, void* out, hipStream_t stream);
hipError_t msr_tile_rows(const float* emb, const int32_t* tile_row, const int32_t* tile_trow, int n_tiles, void* emb_tiled,
                         hipStream_t stream);
hipError_t msr_f16_row_error(const float* emb, const float* inv_norm, int64_t n_rows, uint32_t* err_max, hipStream_t stream);
void msr_gemm_f32_set_dbg(int v);   // honoured by -DMSR_DIAG builds only
hipError_t msr_pad_inv_norm(const float* inv, int64_t n, int64_t n_pad, float* out, hipStream_t stream);
// Top-k of up to 128 normalised queries qn in one pass over the f32 rows; out_n[q] = -1 and *gate |= 1 for a query whose
// entries overflowed (caller: rerun the batch on the sweeps, gated on *gate).  ev (nullable): events around the sample
// pass (0, 1) and the emit pass (2, 3).
hipError_t msr_gemm_f32_topk(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k,
                             int32_t* out_doc, float* out_score, int32_t* out_chunk,
                             int32_t* out_n, int32_t* gate, hipEvent_t* ev, int* width_out, hipStream_t stream);
// The same in two halves, for a doc-sharded index (msr_dense_topk_begin / _end).  _pass: everything up to the thresholds of
// this shard's own tile maxima; out_part[q] (nullable) <- (k_part-th largest tile maximum of query q) - margin[q] / 2, -inf
// when the shard has fewer than k_part row tiles: k_part documents of this shard have EXACT cosines >= that value (a filter
// score is within eps = margin / 2 - 5e-5 of the exact cosine).  _finish: bound (nullable, [nq]): a lower bound A of the
// exact k-th cosine over ALL shards (the minimum over the shards of their out_part with k_part = ceil(k / shards)); a
// document of the global top-k has exact cosine >= A, so filter score >= A - eps: entries below bound - margin / 2 are dropped before
// bucketing, candidate lists and exact rescoring -- the shard then returns fewer than k documents, all it can contribute.
hipError_t msr_gemm_f32_pass(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k, int k_part,
                             float* out_part, hipEvent_t* ev, int* width_out, hipStream_t stream);
hipError_t msr_gemm_f32_finish(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k, const float* bound,
                               int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, int32_t* gate,
                               hipStream_t stream);

// ---- K5: batched bf16 candidate scan finished exactly in f32 (msr_batch.hip) -----------------------
// exact f32 rescoring + final sort of candidate lists that are already filled (cand_n zeroed on return); the candidates name
// the rows to rescore (row j of candidate slot: rows[(q * row_cap + first + j) * row_stride], first | len << 13 in
// cand_chunk[q][slot] on entry; cand_chunk <- the arg-max row): the finish of both streaming passes (f32 rows, bf16 image).
hipError_t msr_batch_rescore_rows(const DenseIndex& ix, const float* qn, int nq, int k, const int32_t* rows, int row_stride,
                                  int row_cap, int32_t* cand_doc, float* cand_score, int32_t* cand_chunk, int32_t* cand_n,
                                  int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, hipStream_t stream);
hipError_t msr_batch_finish(const DenseIndex& ix, const float* qn, int nq, int k, int max_chunks, const float* margin,
                            const float* scores, const float* top_score, const int32_t* top_n, int32_t* cand_doc,
                            float* cand_score, int32_t* cand_chunk, int32_t* cand_n, int32_t* out_doc,
                            float* out_score, int32_t* out_chunk, int32_t* out_n, hipStream_t stream);
