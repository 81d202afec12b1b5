// K5 support -- exact finish of the bf16 candidate scan (gfx950).
//
// The candidate generators (the K-split sweep of msr_dense_ks.hip for <= 128 queries, the tiled GEMM of msr_gemm.hip
// above that) multiply bf16 images e^, q^ of the UNIT vectors e, q and give every document an APPROXIMATE max-cosine s^:
//     |s^ - s| <= eps_q = ||e^ - e|| ||q^|| + ||e|| ||q^ - q|| (+ f32 accumulation)   (Cauchy-Schwarz, twice)
// The two rounding-error norms are MEASURED, not assumed: max_r ||e^_r - e_r|| when the image is built
// (unit_bf16_rows_kernel), ||q^ - q|| per query (batch_margin_kernel); on typical data eps_q is about half of the worst
// case 2^-7, which more than halves the candidate sets.  Let t = k-th largest s^.  The k documents with the largest s^
// have exact scores >= t - eps, so the exact k-th score sigma >= t - eps, and every document of the exact top-k has
// s^ >= sigma - eps >= t - 2 eps.  Hence
//     candidates = { d : s^_d >= t - 2 eps_q }          (margin = 2 eps_q + slack, msr_batch_margin)
// is a superset of the exact top-k.  The kernels below compact that set, recompute the candidates' cosines in
// f32 from the f32 rows (reranker_api.py:285 arithmetic), and sort them exactly.  If a query has more than
// MSR_SEL_CAP candidates the call reports out_n = -1 for it and the host reruns it on the f32 scan.
#include "msr_common.h"
#include "msr_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BT_THREADS = 256;
constexpr int BT_STAGE = 1024;

__global__ __launch_bounds__(BT_THREADS) void thr_compact_kernel(const float* __restrict__ scores, int64_t n,
                                                                  int64_t stride,
                                                                  const float* __restrict__ top_score,
                                                                  const int32_t* __restrict__ top_n, int k,
                                                                  const float* __restrict__ margin,
                                                                  int32_t* __restrict__ cand_doc,
                                                                  int32_t* __restrict__ cand_n) {
    __shared__ int32_t s_doc[BT_STAGE];
    __shared__ int s_n, s_base;
    const int q = blockIdx.y;
    const int have = top_n[q];
    const float thr = have >= k ? top_score[(int64_t)q * k + (k - 1)] - margin[q] : -__builtin_inff();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    const float* row = scores + (int64_t)q * stride;
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * BT_THREADS) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + (int64_t)u * BT_THREADS;
            v[u] = i < hi ? row[i] : -__builtin_inff();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!(msr_valid(v[u]) && v[u] >= thr)) continue;
            const int32_t d = (int32_t)(i0 + (int64_t)u * BT_THREADS);
            const int pos = atomicAdd(&s_n, 1);
            if (pos < BT_STAGE) {
                s_doc[pos] = d;
            } else {
                const int g = atomicAdd(&cand_n[q], 1);
                if (g < MSR_SEL_CAP) cand_doc[(int64_t)q * MSR_SEL_CAP + g] = d;
            }
        }
    }
    __syncthreads();
    int cnt = s_n < BT_STAGE ? s_n : BT_STAGE;
    if (cnt == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&cand_n[q], cnt);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += BT_THREADS)
        if (s_base + j < MSR_SEL_CAP) cand_doc[(int64_t)q * MSR_SEL_CAP + s_base + j] = s_doc[j];
}

// Workgroups per query of the rescoring kernel (4 waves each, a wave takes candidates slot, slot + waves, ...): a few hundred
// candidates at most, and with thousands of queries per call launching 64 mostly idle workgroups per query costs more than the
// rescoring (65 k workgroups per 1024 queries: ~0.2 ms of dispatch for ~0.06 ms of work on a shard of an 8-way run).
static unsigned rescore_grid_x(int nq) { return nq >= 1024 ? 8u : nq >= 512 ? 16u : nq >= 128 ? 32u : 64u; }

// One wave per (query, candidate): exact f32 max-cosine over the document's chunks and its first arg-max.
__global__ __launch_bounds__(BT_THREADS) void rescore_kernel(DenseIndex ix, const float* __restrict__ qn,
                                                              int max_chunks, const int32_t* __restrict__ cand_doc,
                                                              const int32_t* __restrict__ cand_n,
                                                              float* __restrict__ cand_score,
                                                              int32_t* __restrict__ cand_chunk) {
    const int q = blockIdx.y, lane = threadIdx.x & 63;
    int cnt = cand_n[q];
    if (cnt > MSR_SEL_CAP) cnt = MSR_SEL_CAP;
    // (a wave without a candidate leaves before it loads the query: with a node-wide bound a shard keeps ~20 candidates per
    // query, and 2048 queries x 256 waves each fetching 3 KB of query for nothing were most of this kernel's time)
    if ((int)(blockIdx.x * (BT_THREADS / 64) + (threadIdx.x >> 6)) >= cnt) return;
    const f32x4* q4 = (const f32x4*)(qn + (size_t)q * MSR_DIM);
    const f32x4 qa = q4[lane], qb = q4[lane + 64], qc = q4[lane + 128];
    for (int slot = blockIdx.x * (BT_THREADS / 64) + (threadIdx.x >> 6); slot < cnt;
         slot += gridDim.x * (BT_THREADS / 64)) {
        const int d = cand_doc[(int64_t)q * MSR_SEL_CAP + slot];
        const int64_t ds = ix.doc_off[d];
        int64_t de = ix.doc_off[d + 1];
        if (max_chunks > 0 && ds + max_chunks < de) de = ds + max_chunks;
        float best = -__builtin_inff();
        int64_t arg = -1;
        for (int64_t c = ds; c < de; ++c) {
            const f32x4* p = (const f32x4*)(ix.emb + (size_t)c * MSR_DIM);
            const f32x4 a = p[lane], b = p[lane + 64], e = p[lane + 128];
            float s = a.x * qa.x + a.y * qa.y + a.z * qa.z + a.w * qa.w;
            s += b.x * qb.x + b.y * qb.y + b.z * qb.z + b.w * qb.w;
            s += e.x * qc.x + e.y * qc.y + e.z * qc.z + e.w * qc.w;
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            s *= ix.inv_norm[c];
            if (s > best) { best = s; arg = c; }
        }
        if (lane == 0) {
            cand_score[(int64_t)q * MSR_SEL_CAP + slot] = best;
            cand_chunk[(int64_t)q * MSR_SEL_CAP + slot] = (int32_t)arg;
        }
    }
}

// The same for candidates that come with the rows to look at (the f32 streaming pass: gemm_f32_cand_kernel): candidate slot
// of query q owns rows[(q * row_cap + first + j) * row_stride], j < len, with first | len << 13 in cand_chunk[q][slot] on
// entry -- a document's rows in any order (the f32 pass hands them over in descending order and they are walked from the
// back; the bf16 pass orders them by approximate score): the arg-max is the LOWEST row among equal maxima, as above.
// Same arithmetic per row as rescore_kernel.  cand_chunk[q][slot] <- the arg-max row.
__global__ __launch_bounds__(BT_THREADS) void rescore_rows_kernel(DenseIndex ix, const float* __restrict__ qn,
                                                                   const int32_t* __restrict__ rows, int row_stride, int row_cap,
                                                                   const int32_t* __restrict__ cand_n,
                                                                   float* __restrict__ cand_score,
                                                                   int32_t* __restrict__ cand_chunk) {
    const int q = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = cand_n[q];
    if (cnt > MSR_SEL_CAP) return;                      // (overflow: no runs were written; rescore_final_kernel reports -1)
    if ((int)(blockIdx.x * (BT_THREADS / 64) + (threadIdx.x >> 6)) >= cnt) return;
    const f32x4* q4 = (const f32x4*)(qn + (size_t)q * MSR_DIM);
    const f32x4 qa = q4[lane], qb = q4[lane + 64], qc = q4[lane + 128];
    for (int slot = blockIdx.x * (BT_THREADS / 64) + (threadIdx.x >> 6); slot < cnt;
         slot += gridDim.x * (BT_THREADS / 64)) {
        const int fl = cand_chunk[(int64_t)q * MSR_SEL_CAP + slot];
        const int first = fl & 8191;
        int len = fl >> 13;
        if (first + len > row_cap) len = row_cap - first;
        float best = -__builtin_inff();
        int64_t arg = -1;
        for (int j = len - 1; j >= 0; --j) {
            const int64_t c = rows[((int64_t)q * row_cap + first + j) * row_stride];
            if (c < 0 || c >= ix.n_chunks) continue;    // (validated by the producer; never trusted as an address)
            const f32x4* p = (const f32x4*)(ix.emb + (size_t)c * MSR_DIM);
            const f32x4 a = p[lane], b = p[lane + 64], e = p[lane + 128];
            float s = a.x * qa.x + a.y * qa.y + a.z * qa.z + a.w * qa.w;
            s += b.x * qb.x + b.y * qb.y + b.z * qb.z + b.w * qb.w;
            s += e.x * qc.x + e.y * qc.y + e.z * qc.z + e.w * qc.w;
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            s *= ix.inv_norm[c];
            if (s > best || (s == best && c < arg)) { best = s; arg = c; }   // (the FIRST arg-max, whatever the order of the run)
        }
        if (lane == 0) {
            cand_score[(int64_t)q * MSR_SEL_CAP + slot] = best;
            cand_chunk[(int64_t)q * MSR_SEL_CAP + slot] = (int32_t)arg;
        }
    }
}

// One workgroup per query: sort the rescored candidates by (score desc, doc asc) and emit the top-k.
__global__ __launch_bounds__(1024) void rescore_final_kernel(const int32_t* __restrict__ cand_doc,
                                                              const float* __restrict__ cand_score,
                                                              const int32_t* __restrict__ cand_chunk,
                                                              int32_t* __restrict__ cand_n, int k,
                                                              int32_t* __restrict__ out_doc,
                                                              float* __restrict__ out_score,
                                                              int32_t* __restrict__ out_chunk,
                                                              int32_t* __restrict__ out_n) {
    __shared__ uint64_t key[MSR_SEL_CAP];           // (orderable(score) << 32) | ~doc : all distinct
    __shared__ uint32_t val[MSR_SEL_CAP];           // candidate slot
    const int q = blockIdx.x, t = threadIdx.x;
    const int raw = cand_n[q];
    const bool overflow = raw > MSR_SEL_CAP;
    const int cnt = overflow ? MSR_SEL_CAP : raw;
    int P = 64;
    while (P < cnt) P <<= 1;
    for (int i = t; i < P; i += 1024) {
        uint64_t kk = 0;
        if (i < cnt) {
            const float s = cand_score[(int64_t)q * MSR_SEL_CAP + i];
            if (msr_valid(s))
                kk = ((uint64_t)msr_ord32(s) << 32) | (uint32_t)~(uint32_t)cand_doc[(int64_t)q * MSR_SEL_CAP + i];
        }
        key[i] = kk; val[i] = (uint32_t)i;
    }
    __syncthreads();
    for (int kk = 2; kk <= P; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int idx = t; idx < (P >> 1); idx += 1024) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int p = i | j;
                const bool desc = (i & kk) == 0;
                const uint64_t a = key[i], b = key[p];
                if (desc ? a < b : a > b) {
                    key[i] = b; key[p] = a;
                    const uint32_t v = val[i]; val[i] = val[p]; val[p] = v;
                }
            }
            __syncthreads();
        }
    }
    const int n_sel = overflow ? 0 : (cnt < k ? cnt : k);
    for (int i = t; i < k; i += 1024) {
        const int64_t o = (int64_t)q * k + i;
        const bool ok = i < n_sel && key[i] != 0;
        out_doc[o] = ok ? (int32_t)~(uint32_t)key[i] : -1;
        out_score[o] = ok ? msr_unord32((uint32_t)(key[i] >> 32)) : -__builtin_inff();
        if (out_chunk) out_chunk[o] = ok ? cand_chunk[(int64_t)q * MSR_SEL_CAP + val[i]] : -1;
    }
    if (t == 0) {
        out_n[q] = overflow ? -1 : n_sel;            // -1: rerun this query on the exact f32 scan
        cand_n[q] = 0;
    }
}

}  // namespace

hipError_t msr_batch_rescore_rows(const DenseIndex& ix, const float* qn, int nq, int k, const int32_t* rows, int row_stride,
                                  int row_cap, int32_t* cand_doc, float* cand_score, int32_t* cand_chunk, int32_t* cand_n,
                                  int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if (row_cap > 8192 || row_stride < 1) return hipErrorInvalidValue;
    rescore_rows_kernel<<<dim3(rescore_grid_x(nq), (unsigned)nq), BT_THREADS, 0, stream>>>(ix, qn, rows, row_stride, row_cap, cand_n,
                                                                                          cand_score, cand_chunk);
    rescore_final_kernel<<<nq, 1024, 0, stream>>>(cand_doc, cand_score, cand_chunk, cand_n, k, out_doc, out_score,
                                                  out_chunk, out_n);
    return hipGetLastError();
}

hipError_t msr_batch_finish(const DenseIndex& ix, const float* qn, int nq, int k, int max_chunks, const float* margin,
                            const float* scores, const float* top_score, const int32_t* top_n, int32_t* cand_doc,
                            float* cand_score, int32_t* cand_chunk, int32_t* cand_n, int32_t* out_doc,
                            float* out_score, int32_t* out_chunk, int32_t* out_n, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    int64_t parts = (ix.n_docs + 8191) / 8192;
    const int64_t max_parts = 2048 / nq > 0 ? 2048 / nq : 1;
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    thr_compact_kernel<<<dim3((unsigned)parts, (unsigned)nq), BT_THREADS, 0, stream>>>(scores, ix.n_docs, ix.score_stride, top_score,
                                                                                      top_n, k, margin, cand_doc, cand_n);
    rescore_kernel<<<dim3(rescore_grid_x(nq), (unsigned)nq), BT_THREADS, 0, stream>>>(ix, qn, max_chunks, cand_doc, cand_n, cand_score,
                                                                    cand_chunk);
    rescore_final_kernel<<<nq, 1024, 0, stream>>>(cand_doc, cand_score, cand_chunk, cand_n, k, out_doc, out_score,
                                                  out_chunk, out_n);
    return hipGetLastError();
}
