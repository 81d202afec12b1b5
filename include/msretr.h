/*
 * msretr.h -- C ABI of the MI355X-native two-stage retriever (libmsretr.so).
 *
 * The reference (StephenTaf/Modern-Search-Engines-Project) has no FFI / plugin interface: its query path
 * is three Python call sites.  Each entry point below replaces one of them and is what a binding for that
 * call site would bind (see INTEGRATION.md for the ctypes stubs):
 *
 *   msr_bm25_topk   <- BM25.search scoring loop + sort + cut      indexer/bm25_indexer.py:434-488
 *   msr_dense_topk  <- Retriever.quick_search (dense full scan)   search_api.py:60,87 (retriever.py absent;
 *                      cosine reranker/reranker_api.py:285, per-doc max :370, report p.2)
 *   msr_rerank      <- /rerank endpoint arithmetic                reranker/reranker_api.py:27-63,273-334,357-372
 *   msr_merge_topk  <- (new) merge of per-shard top-k after the RCCL all-gather; no reference counterpart
 *   msr_bind_*      <- the per-query SQL fetches, done ONCE        bm25_indexer.py:413-448, reranker_api.py:36-61
 *
 * Conventions
 *   - Every function returns 0 on success or a negative msr_status; it never throws and never returns
 *     memory the caller has to free.  msr_last_error() gives the text of the last failure.
 *   - All array arguments are DEVICE pointers into HBM of the engine's device unless marked [host].
 *     The caller owns them (typically torch tensors: tensor.data_ptr()) and must keep the bound index
 *     arrays alive until msr_destroy / the next msr_bind_*.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).  Work is only
 *     enqueued; nothing synchronises.  Results are valid once the stream has drained.
 *   - A document is addressed by its dense index: the rank of its doc_id in ascending order, so
 *     "ascending index" == "ascending doc_id" (the reference's tie order, bm25_indexer.py:445,484).
 *     A shard adds `doc_base` to its local indices before the merge.
 *   - One engine per (device, stream); an engine is not thread-safe, distinct engines are independent.
 */
#ifndef MSRETR_H
#define MSRETR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSR_ABI_VERSION 3
#define MSR_DIM 768               /* config.py:2 EMBEDDING_DIMENSION */
#define MSR_MAX_K 1024            /* config.py:13 TOP_K_RETRIEVAL = 1000 */
#define MSR_MAX_QUERY_TERMS 64
#define MSR_RERANK_MAX_CHUNKS 10  /* reranker_api.py:58 */

typedef enum msr_status {
    MSR_OK = 0,
    MSR_ERR_INVALID = -1,    /* bad argument (null pointer, size out of range, wrong dim) */
    MSR_ERR_NOT_BOUND = -2,  /* the index part this call needs has not been bound */
    MSR_ERR_HIP = -3,        /* a HIP runtime call failed; text in msr_last_error */
    MSR_ERR_NOMEM = -4       /* scratch allocation failed */
} msr_status;

typedef struct msr_engine msr_engine;

typedef struct msr_config {
    int32_t struct_size;      /* sizeof(msr_config), for forward compatibility */
    int32_t device;           /* HIP device ordinal */
    int32_t dim;              /* must be MSR_DIM */
    int32_t max_queries;      /* queries processed per internal slice; scratch is sized for this many */
    int32_t max_k;            /* largest k any call will ask for, <= MSR_MAX_K */
    int32_t rerank_max_docs;  /* largest candidate list per query for msr_rerank, <= 1024 */
    int32_t scan_layout;      /* 0 = row-major embeddings; 1 = 16-row interleaved */
    int32_t scan_variant;     /* 0 = default: f16-split products when every row norm is in [0.5, 2], else exact f32;
                                 2 = always the exact-f32 MFMA kernel; 7 = f16-split on the 32-query kernel; 14 = the K-split kernel
                                 (what 0 resolves to on unit-norm rows); 15 = the K-split kernel over a pre-split f16 hi/lo copy of
                                 the rows (+4 bytes per value of HBM, ~4 % faster).  Any other value: msr_create fails */
    int32_t flags;            /* MSR_CFG_* bits; unknown bits: msr_create fails (ABI 3) */
} msr_config;

/* msr_config.flags */
#define MSR_CFG_NO_ROW_COPY 1 /* do not build the fragment-order copy of the embedding matrix (msr_bind_chunks): the 256-query
                                 pass then reads the caller's row-major matrix (same results bit for bit, ~13 % slower pass,
                                 half the embedding footprint); also declines the f16 image of the rows that launches of
                                 several query groups read (max_queries >= 512; msr_row_image_state) */

/* BM25 parameters travel with the postings (bm25_indexer.py:57 k1=1.2, b=0.75). */
typedef struct msr_rerank_params {
    double smoothing;          /* reranker/config.yaml:28   0.15 */
    double max_boost;          /* reranker_api.py:317       0.10 */
    double max_decay;          /* reranker_api.py:318       0.05 */
    int32_t max_chunks;        /* reranker_api.py:58        10   */
    int32_t reserved;
} msr_rerank_params;

int msr_abi_version(void);
int msr_create(const msr_config* cfg, msr_engine** out);
int msr_destroy(msr_engine* e);
/* e may be NULL: returns the text of the last failed msr_create on this thread. */
const char* msr_last_error(const msr_engine* e);

/* Stage-1 index: CSR postings over the dense doc index, sorted by doc inside each term.
 *   term_off[n_terms+1] i64, post_doc[n_postings] i32, post_tf[n_postings] i32   bm25_term_freq  (:97-104)
 *   doc_len[n_docs] i32                                                          bm25_doc_stats  (:88-94)
 *   idf[n_terms] f32 (as stored: REAL, may be negative)                          bm25_term_stats (:106-113)
 *   avgdl f32 (REAL)                                                             bm25_corpus_stats (:116-122)
 */
int msr_bind_postings(msr_engine* e, const int64_t* term_off, int64_t n_terms, const int32_t* post_doc,
                      const int32_t* post_tf, int64_t n_postings, const int32_t* doc_len, int64_t n_docs,
                      const float* idf, float avgdl, double k1, double b, void* stream);

/* Stage-2 index: chunk embeddings sorted by (doc index, chunk_id); doc d owns rows
 * [doc_off[d], doc_off[d+1]).  emb is f32 [n_chunks][768] row-major (scan_layout 0) or the interleaved
 * image produced by msr_interleave_rows (scan_layout 1).  inv_norm[n_chunks] f32 = 1/||row|| (0-norm -> 1),
 * or NULL to have the engine compute it.       indexer/embedder.py:31-52, indexer/indexer.py:165
 * Engine-owned memory this call allocates besides small tables: when cfg.max_queries >= 256 (and the corpus qualifies for
 * the streaming pass: row-major, documents of <= 256 chunks, >= 64 row tiles) a copy of the matrix in the order the
 * 256-query pass loads it, 1.03 x n_chunks x 3072 bytes; the caller's matrix stays bound and is what
 * every other kernel reads, so it must stay alive.  SURVEY 8b's "the handle owns only small scratch" is deliberately not
 * kept here: the copy is a trade of HBM for bandwidth, with a switch (MSR_CFG_NO_ROW_COPY) and a fallback (msr_row_copy_state).
 * When cfg.max_queries >= 512 (calls that put several 256-query groups into one launch: those launches are bound by the
 * matrix pipes, not by HBM) also an f16 image of the rows, (n_chunks + 512) x 1536 bytes: the values the 256-query pass
 * otherwise converts in registers, group after group -- the same products, the same results bit for bit.  Launches of ONE
 * group (<= 256 queries per call: the HBM-bound case) never read it; same switch, same fallback (msr_row_image_state). */
int msr_bind_chunks(msr_engine* e, const float* emb, int64_t n_chunks, const int32_t* doc_off,
                    int64_t n_docs, const float* inv_norm, void* stream);
/* What became of that copy: 0 = not applicable (max_queries < 256 or the corpus does not qualify), 1 = built and used,
 * 2 = declined by MSR_CFG_NO_ROW_COPY, 3 = its allocation failed and the engine fell back to the row-major matrix (the bind
 * still succeeds).  -1: null handle. */
int msr_row_copy_state(const msr_engine* e);
/* The same for the f16 image of the rows that launches of several 256-query groups read: 0 = not applicable (max_queries < 512
 * or the corpus does not qualify), 1 = built and used, 2 = declined by MSR_CFG_NO_ROW_COPY, 3 = its allocation failed (those
 * launches then convert the f32 rows in registers as a single-group launch does).  -1: null handle. */
int msr_row_image_state(const msr_engine* e);
/* Device memory the handle owns right now, in bytes (scratch, tables built at bind, the copies above); the caller's bound
 * arrays are not included. */
int64_t msr_owned_bytes(const msr_engine* e);

/* Per-document metadata the rerank stage needs: url_group[n_docs] i32 = id of the document's URL with
 * the query string removed, or -1 when the document is not in urlsDB.   reranker_api.py:38-47 */
int msr_bind_doc_meta(msr_engine* e, const int32_t* url_group, int64_t n_docs, void* stream);

/* Arithmetic of the bound dense scan's SWEEPS (calls of <= 64 queries, and the fallback): 0 = exact f32 MFMA (bit-for-bit
 * a k-ordered fmaf chain), 1 = f32 rows split into two f16 pieces, three f16 MFMAs per k-step with f32 accumulation
 * (|error| <= 8e-6 on the cosine for row norms in [0.5, 2], proof in DESIGN.md); -1 = no chunks bound.  When
 * msr_scan_width() says 128 or 256, calls of more than 64 queries take one streaming pass per 128 (256) queries instead:
 * an f16 filter with a measured margin, then EXACT f32 cosines for every returned document. */
int msr_scan_arith(const msr_engine* e);

/* Most queries one pass over the embedding matrix serves in msr_dense_topk: 256 / 128 when the streaming pass is available
 * (f16-split arithmetic bound, every document within one 256-row tile, enough tiles; 256 needs max_queries >= 256 and a
 * call of more than 128 queries), 64 when only the K-split kernel is (row-major layout, no per-document row limit, a
 * corpus that meets its preconditions; both arithmetics), else 32; -1 = no chunks bound. */
int msr_scan_width(const msr_engine* e);
/* Queries per pass over the matrix of the kernel the MOST RECENT msr_dense_topk call ran (256 / 128: the streaming pass;
 * 64 / 32: the sweeps -- also what a call takes whose k or max_chunks_per_doc the streaming pass does not serve); 0 before
 * the first call.  For whoever attributes a measured kernel time to a kernel (bench.py). */
int msr_dense_path(const msr_engine* e);
/* The same for the <= 128-query sweeps of msr_dense_topk_bf16 (after msr_enable_bf16): 128, 64, or -1. */
int msr_batch_width(const msr_engine* e);
/* 1 if calls of msr_dense_topk_bf16 with more than 128 queries run as the tiled matrix-core GEMM (every document fits a
 * 256-row tile and the corpus has enough tiles), else 0 (such calls are served by repeated sweeps). */
int msr_batch_gemm_ok(const msr_engine* e);

/* Re-order row-major rows into the 16-row interleaved scan layout (dst may not alias src).
 * n_rows is padded up to a multiple of 16 in dst (pad rows zero): dst holds ceil16(n_rows)*768 floats. */
int msr_interleave_rows(msr_engine* e, const float* src, int64_t n_rows, float* dst, void* stream);

/* BM25 top-k for Q queries.  Query q owns q_terms/q_qtf[q_term_off[q] .. q_term_off[q+1]): its UNIQUE
 * term ids in first-occurrence order and how often each occurs in the query (bm25_indexer.py:405-409);
 * ids outside [0, n_terms) or with empty posting lists are skipped (:430).  Scores are float64 and
 * bit-identical to the reference's accumulation order (:466-478).  A document is a candidate only if at
 * least one posting touched it and score >= min_score (:461,480).  Output row q holds out_n[q] <= k
 * entries ordered by (score desc, doc index asc) (:484-485); the rest of the row is -1 / -inf. */
int msr_bm25_topk(msr_engine* e, const int32_t* q_term_off, const int32_t* q_terms, const int32_t* q_qtf,
                  int32_t n_queries, int32_t k, double min_score, int32_t* out_doc, double* out_score,
                  int32_t* out_n, void* stream);

/* Dense full scan for Q queries: score(d) = max over the document's first `max_chunks_per_doc` chunks
 * (0 = all) of cosine(q, chunk), cosine as sklearn computes it in float32 (reranker_api.py:285), to within the
 * 1e-5 tolerance of the task (see msr_scan_arith for the arithmetic actually used).
 * q is [n_queries][768] f32, NOT normalised (reranker_api.py:355).  Output rows as for msr_bm25_topk,
 * with float32 scores and out_chunk = row index of the arg-max chunk (first maximum). */
int msr_dense_topk(msr_engine* e, const float* q, int32_t n_queries, int32_t k, int32_t max_chunks_per_doc,
                   int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, void* stream);

/* msr_dense_topk in two halves, for a doc-sharded index: between them the caller exchanges ONE float per query across the
 * shards (msretr/distributed.py: an all-reduce MIN over RCCL), after which every shard rescores only the documents that can
 * be in the top-k of the WHOLE corpus instead of its own top-k -- 1 / shards of the exact-f32 rescoring per rank.
 *   msr_dense_split_max(e, k): most queries one begin / end pair takes (0: this engine cannot split -- corpus without row
 *     tiles, fewer than 2 k tiles -- use msr_dense_topk).  Pairs take MORE than 64 queries.
 *   msr_dense_topk_begin: the passes over this shard's rows and the thresholds of its own tile maxima.  out_part[q] f32
 *     [n_queries] (device) <- a cosine that k_part documents of THIS shard are guaranteed to reach EXACTLY (their filter
 *     scores minus the measured error margin); -inf when the shard has fewer than k_part row tiles.  With
 *     k_part = ceil(k / shards), the minimum of out_part[q] over all shards is a lower bound of the k-th exact cosine of the
 *     whole corpus: shards x k_part >= k documents reach it.
 *   msr_dense_topk_end: bound [n_queries] (device; NULL: none) = that minimum.  Candidates whose filter score lies below
 *     bound - (half the filter's measured error margin) cannot be in the global top-k and are dropped before the candidate lists and the exact rescoring.  Output
 *     as msr_dense_topk, except that out_n[q] may be < k: the shard returns every document it can contribute to the global
 *     top-k (merge the shards' lists with msr_merge_topk_payload as usual; the merged list is the unsharded one, bit for bit).
 * One begin may be pending per engine; the matching end must follow with the same n_queries and k. */
int msr_dense_split_max(const msr_engine* e, int32_t k);
int msr_dense_topk_begin(msr_engine* e, const float* q, int32_t n_queries, int32_t k, int32_t k_part, float* out_part,
                         void* stream);
int msr_dense_topk_end(msr_engine* e, int32_t n_queries, int32_t k, const float* bound, int32_t* out_doc, float* out_score,
                       int32_t* out_chunk, int32_t* out_n, void* stream);

/* Batched variant of msr_dense_topk for throughput (BASELINE config 5).  Candidates come from a bf16 image of the
 * NORMALISED rows (v_mfma_f32_16x16x32_bf16, f32 accumulation), whose scores carry a proven error bound; every document
 * within twice that bound of the k-th approximate score is re-scored in f32 from the f32 rows, so the final top-k is
 * exact (same ordering rule as msr_dense_topk).  Up to 128 queries per call: one sweep of the image (msr_batch_width).
 * More: a tiled GEMM, 1024 queries per pass over the image, that never writes the score matrix (msr_batch_gemm_ok; design
 * in csrc/msr_gemm.hip).  A query whose candidate set exceeds the engine's capacity comes back with out_n = -1: rerun it
 * with msr_dense_topk.  msr_enable_bf16 builds the image (+2 bytes per embedding value of HBM) and the scratch of the
 * GEMM path (~0.8 GB); row-major layout only. */
int msr_enable_bf16(msr_engine* e, void* stream);
int msr_dense_topk_bf16(msr_engine* e, const float* q, int32_t n_queries, int32_t k, int32_t max_chunks_per_doc,
                        int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, void* stream);

/* Rerank/fuse of stage-1 candidates, reranker_api.py:337-372.  Query qi has cand_n[qi] <= max_cand
 * candidates in row qi of cand_doc / cand_bm25 (any order).  Candidates not in urlsDB, losing the URL
 * dedup (MIN(id) wins) or without chunks are dropped; the others come back ordered by
 * (new_similarity desc, doc index asc): out_doc, out_score (new_similarity), out_orig (min-max
 * normalised BM25 of the winning row), out_chunk (row index of the winning chunk), out_n, and
 * out_rows[qi] = number of chunk rows that took part (RerankResponse.total_documents, :410). */
int msr_rerank(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
               const double* cand_bm25, const int32_t* cand_n, int32_t max_cand,
               const msr_rerank_params* params, int32_t* out_doc, double* out_score, double* out_orig,
               int32_t* out_chunk, int32_t* out_n, int32_t* out_rows, void* stream);

/* Domain diversification of fused lists on the device (reranker_api.py:170-236 hybrid_diversification; :376-397 the response
 * models rejecting NULL title / url / text), so that a batch brings back top_k rows per query instead of max_cand.
 *   msr_bind_doc_domains: domain[d] i32 for every document index the fused lists may hold (GLOBAL indices): the id of
 *     urlparse(url).netloc.lower() (any numbering; equal ids = same domain), or -1 for a document that never appears in a
 *     response.  Borrowed like the other bound arrays; NULL unbinds (every document accepted, each its own domain).
 *   msr_diversify: fused_* [n_queries][max_cand] / fused_n [n_queries] as msr_rerank / msr_rerank_fuse return them (ordered by
 *     (new_similarity desc, doc asc)).  diversify != 0: one result per domain among the domains whose best entry scores
 *     >= relevance_threshold (0.8), then one per remaining domain up to top_k, then -- if still short of top_k -- the dropped
 *     entries with their scores shifted below the last kept one (delta = first_dropped - last_kept + 1e-4, clamped at 0),
 *     float64, the reference's operations in the reference's order.  diversify == 0: the first top_k accepted entries.
 *     out_* [n_queries][max_cand] (like the reference, the list may exceed top_k when more than top_k domains are "high");
 *     rows past out_n[q] are -1 / -inf. */
int msr_bind_doc_domains(msr_engine* e, const int32_t* domain, int64_t n_docs, void* stream);
int msr_diversify(msr_engine* e, int32_t n_queries, const int32_t* fused_doc, const double* fused_score,
                  const double* fused_orig, const int32_t* fused_chunk, const int32_t* fused_n, int32_t max_cand,
                  int32_t top_k, double relevance_threshold, int32_t diversify, int32_t* out_doc, double* out_score,
                  double* out_orig, int32_t* out_chunk, int32_t* out_n, void* stream);

/* HOST function (every pointer is host memory; no device is touched): the batch result lines of search_api.py:290,
 * "{query_num}\t{rank}\t{url}\t{score:.3f}\n", for n_queries final lists in one call.  Query q's number is the bytes
 * qnum_blob[qnum_off[q] .. qnum_off[q+1]); its list is doc / score [q * stride .. + n[q]) (rank = position + 1); the URL of
 * document d is url_blob[url_off[d] .. url_off[d+1]) (UTF-8; d outside [0, n_docs): empty); max_url_len = the longest URL in
 * bytes (sizes the output without touching the table; <= 0: computed here, one pass over url_off).  The score is printed exactly as
 * Python's format(score, ".3f").  Returns the bytes written; if `capacity` is below the function's upper bound of them nothing
 * is written and the result is -(that bound) (call with capacity 0 to size the buffer); INT64_MIN for a bad argument. */
int64_t msr_format_lines(const char* qnum_blob, const int64_t* qnum_off, int32_t n_queries, const int32_t* doc,
                         const double* score, const int32_t* n, int32_t stride, const char* url_blob, const int64_t* url_off,
                         int64_t n_docs, int64_t max_url_len, char* out, int64_t capacity);

/* The two halves of msr_rerank, for a doc-sharded index (the reference-exact hybrid needs a
 * second exchange).  cand_doc holds GLOBAL document indices and is identical on every rank.
 *   msr_rerank_gather: for the candidates this shard owns (doc_base <= doc < doc_base + n_docs) writes the
 *     cosines of their first <= max_chunks chunks into out_cos[q][m][0..10) and
 *     out_meta[q][m] = (rows, url_group + 2, row_base + first row); everything else is written as 0.
 *     Summing out_cos / out_meta over the shards (one RCCL all-reduce, integer SUM of the raw bits,) yields the arrays of the whole
 *     candidate list, because exactly one shard contributes a non-zero entry.
 *   msr_rerank_fuse: the float64 chain of reranker_api.py:360-372 on those arrays; touches no index, so
 *     every rank computes the same result.  Outputs as msr_rerank (out_doc are global indices). */
int msr_rerank_gather(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                      const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base,
                      int32_t max_chunks, float* out_cos, int32_t* out_meta, void* stream);
/* msr_rerank_gather writing straight into the send buffer of the all-to-all that carries every query's halves to the rank
 * that fuses it (msretr/distributed.py): out_blocks is [n_blocks][block_words] int32; block b belongs to the rank that owns
 * queries [b * queries_per_block, (b + 1) * queries_per_block) and holds their cos rows ([queries_per_block][max_cand][10],
 * float bits) followed by their meta rows ([queries_per_block][max_cand][3]); block_words >= queries_per_block * max_cand * 13
 * (any padding is left untouched).  One launch for all queries of the call (cfg.max_queries of them at a time; for larger
 * calls max_queries must be a multiple of queries_per_block). */
int msr_rerank_gather_blocks(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                             const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base,
                             int32_t max_chunks, int32_t* out_blocks, int32_t queries_per_block, int64_t block_words,
                             void* stream);
int msr_rerank_fuse(msr_engine* e, int32_t n_queries, const int32_t* cand_doc, const double* cand_bm25,
                    const int32_t* cand_n, int32_t max_cand, const float* cos, const int32_t* meta,
                    const msr_rerank_params* params, int32_t* out_doc, double* out_score, double* out_orig,
                    int32_t* out_chunk, int32_t* out_n, int32_t* out_rows, void* stream);

/* Join of the per-shard halves of msr_rerank_gather after they have been exchanged (msretr/distributed.py sends every
 * shard's half of a query to the rank that fuses that query: one all-to-all).  Part p holds cos [n_queries][max_cand][10] at
 * cos_parts + p * part_stride_bytes and meta [n_queries][max_cand][3] at meta_parts + p * part_stride_bytes (4-byte aligned;
 * 16-byte aligned pointers and stride take the wide path).  out = bitwise OR over the parts: exactly one shard owns a
 * candidate's document and wrote non-zero words for it, so the OR is that shard's entry.  No reference counterpart
 * (reranker_api.py:27-63 fetches all rows from one database). */
int msr_rerank_combine(msr_engine* e, const float* cos_parts, const int32_t* meta_parts, int32_t n_parts,
                       int64_t part_stride_bytes, int32_t n_queries, int32_t max_cand, float* out_cos,
                       int32_t* out_meta, void* stream);

/* The COMPACT form of that exchange (round 4; msretr/distributed.py uses it by default).  A rank of an N-way run owns ~1/N of a
 * query's candidates, so the blocks of msr_rerank_gather_blocks are mostly zero words (52 KB per query and rank at max_cand =
 * 1000).  Here a rank sends one RECORD of 16 words per candidate slot it owns -- [slot, rows, url_group + 2, first row,
 * cos x 10 (float bits), query, 0] -- and every rank can size the exchange without talking to anyone: the merged candidate
 * lists are replicated and the shards are document ranges (shard s owns shard_bounds[s] <= doc < shard_bounds[s + 1], device
 * array of n_shards + 1), so
 *   msr_rerank_plan counts, for ALL shards, counts[s][q] = candidates of query q that shard s owns, and derives from it
 *     send_base[q] / send_blk[q][ceil(max_cand / 8)]: the record number of the first owned slot of query q / of each block of
 *       8 slots within the query, in THIS rank's send buffer (records ordered by query, then slot: the records for the
 *       queries of rank o -- queries [o * queries_per_shard, (o + 1) * queries_per_shard) -- are contiguous),
 *     recv_off[s][j]: the record number, in the receive buffer, of the first record source s sends for my j-th query,
 *     pair[s][o]: records source s sends to rank o -- the split sizes of the all-to-all (x 16 words), which the host reads;
 *   msr_rerank_gather_records is msr_rerank_gather writing those records (nothing for slots of other shards; out_records
 *     holds capacity_records records -- n_queries * max_cand is always enough -- and a record whose number is not below that
 *     is not written);
 *   msr_rerank_scatter puts the received records (a buffer of capacity_records records: nothing past it is read) of my
 *     queries [first_query, first_query + n_my_queries) into the dense
 *     out_cos [n_my_queries][max_cand][10] / out_meta [..][3] msr_rerank_fuse reads (zeroed first: a slot nobody owns stays
 *     "no document").  The result equals msr_rerank_combine over the dense halves, bit for bit.
 * All arrays are device pointers.  No reference counterpart (reranker_api.py:27-63 fetches all rows from one database). */
int msr_rerank_plan(msr_engine* e, int32_t n_queries, const int32_t* cand_doc, const int32_t* cand_n, int32_t max_cand,
                    const int32_t* shard_bounds, int32_t n_shards, int32_t my_shard, int32_t queries_per_shard,
                    int32_t* counts, int32_t* send_base, int32_t* send_blk, int32_t* recv_off, int32_t* pair, void* stream);
int msr_rerank_gather_records(msr_engine* e, const float* q, int32_t n_queries, const int32_t* cand_doc,
                              const int32_t* cand_n, int32_t max_cand, int32_t doc_base, int32_t row_base,
                              int32_t max_chunks, const int32_t* send_base, const int32_t* send_blk, int32_t* out_records,
                              int64_t capacity_records, void* stream);
int msr_rerank_scatter(msr_engine* e, const int32_t* records, int64_t capacity_records, const int32_t* counts,
                       const int32_t* recv_off, int32_t n_shards, int32_t n_queries, int32_t queries_per_shard,
                       int32_t first_query, int32_t n_my_queries, int32_t max_cand, float* out_cos, int32_t* out_meta,
                       void* stream);

/* Merge n_parts per-shard top-k lists (the payload of the RCCL all-gather) into the global top-k.
 * in_doc [n_parts][n_queries][k] i32 GLOBAL doc indices, in_score same shape (score_bits = 32: f32,
 * 64: f64), in_n [n_parts][n_queries].  Order: score desc, doc index asc -- identical on every rank.
 * Every part must be in that order already (what msr_bm25_topk / msr_dense_topk return, with doc indices made global by
 * adding the shard's base): the kernel merges sorted lists, it does not sort. */
int msr_merge_topk(msr_engine* e, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                   int32_t n_parts, int32_t n_queries, int32_t k, int32_t score_bits, int32_t* out_doc,
                   void* out_score, int32_t* out_n, void* stream);
/* The same with (a) an optional 32-bit payload per entry that travels with it: in_payload [n_parts][n_queries][k] ->
 * out_payload [n_queries][k] (-1 past out_n; both NULL: none) -- the dense lists carry their arg-max chunk row this way, so
 * the per-document arg-max of reranker_api.py:370 survives the merge without a second lookup; (b) part_stride_bytes != 0
 * (a multiple of 8): part p of EVERY input array starts that many bytes after part p - 1, i.e. the arrays are read in place
 * from the receive buffer of ONE all-gather whose per-rank record is [doc | score | n | payload ...]; 0: contiguous arrays. */
int msr_merge_topk_payload(msr_engine* e, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                           const int32_t* in_payload, int32_t n_parts, int64_t part_stride_bytes, int32_t n_queries,
                           int32_t k, int32_t score_bits, int32_t* out_doc, void* out_score, int32_t* out_n,
                           int32_t* out_payload, void* stream);

/* BM25 index build on the GPU (handle-less, OFFLINE: unlike every other entry point this one allocates
 * its own workspace and synchronises the stream).  Replaces the term counting and table writes of BM25.build_index
 * (indexer/bm25_indexer.py:16-54, 203-250; doc_freq :130-147) given pre-tokenised documents: document i (documents in
 * ascending doc_id order, only those with at least one token) owns tok_ids[tok_off[i] .. tok_off[i+1]), term ids in
 * [0, n_terms) -- checked on the device, MSR_ERR_INVALID otherwise.  When capacity >= the number of postings: writes
 * term_off[n_terms + 1] and post_doc / post_tf (CSR by term, documents ascending inside a term, tf = occurrences).
 * *n_postings [host] always receives the number of postings: call once with capacity 0 to size the arrays (that call may
 * return after the counting phase and leave term_off untouched), then again.  All arrays are device pointers. */
int msr_build_postings(const int64_t* tok_off, const int32_t* tok_ids, int64_t n_docs, int32_t n_terms, int64_t* term_off,
                       int32_t* post_doc, int32_t* post_tf, int64_t capacity, int64_t* n_postings, void* stream);

/* Timing hooks for bench.py: while enabled, every launch of the dominant kernels is bracketed by a
 * hipEvent pair recorded on the caller's stream (ring of 256 launches per kernel).  msr_kernel_time_ms
 * blocks on the recorded events and returns the SUM of the launch durations and the number of launches
 * since msr_set_timing(e, 1).  which: 0 = dense scan kernel, 1 = BM25 TAAT kernel, 2 = GEMM emit pass (all row tiles),
 * 3 = GEMM sample pass (every 16th tile). */
int msr_set_timing(msr_engine* e, int32_t enabled);
/* Measurement hook of the DIAGNOSTIC build (libmsretr_diag.so, -DMSR_DIAG: knock-out switches of the GEMM kernels,
 * tools/gemm_check.py --dbg).  The product library knows no key and returns MSR_ERR_INVALID: it has no switches, no
 * environment variables and one implementation per kernel. */
int msr_tune(msr_engine* e, int32_t key, int32_t value);
int msr_kernel_time_ms(msr_engine* e, int32_t which, float* out_ms, int32_t* out_launches);

#ifdef __cplusplus
}
#endif
#endif /* MSRETR_H */
