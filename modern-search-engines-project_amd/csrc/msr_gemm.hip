// K5 -- batched candidate generation as a bf16 GEMM on the matrix cores (gfx950): the scheme around the pass.
//
// BASELINE configs[4]: S = E (C x 768, bf16) . Q^T (768 x nq, bf16), f32 accumulate, for batches of hundreds to
// thousands of queries.  Same mathematics as the cosine of the reference (reranker/reranker_api.py:285) and its
// per-document max (:370) -- but at this batch size the work is a GEMM (7.86 TFLOP per 1024 queries at 5 M rows) and
// the bound is the matrix pipe, not HBM.  The score matrix (5 M x 1024) is NEVER written anywhere:
//
//   pass 1 (sample)  the GEMM over every SS-th row tile; epilogue = per (tile, query) maximum only.  Tiles are cut at
//                    document boundaries, so the k-th largest tile maximum is attained by k DIFFERENT documents:
//                    a valid lower bound t_s <= t of the k-th largest per-document score.
//   pass 2 (emit)    the GEMM over all tiles; epilogue = (a) the same tile maxima for ALL tiles (they give a much
//                    tighter bound t' afterwards), (b) every (query, row, score) with score >= t_s - margin is
//                    appended to a buffer private to the WAVE (scalar counter + lane prefix count, plain 16 B stores:
//                    no atomic of any kind, nothing that would drain the load pipeline).
//   finish           entries >= t' - margin are bucketed per query, reduced to per-document maxima, and the survivors
//                    are re-scored exactly in f32 by the kernels of msr_batch.hip.
//
// Both operands are bf16 images of UNIT vectors (rows are normalised when the image is built).  The margin is MEASURED,
// not a worst case: unit_bf16_rows_kernel records dE = the largest rounding-error norm |e - bf16(e)| over all rows,
// batch_margin_kernel computes dq the same way per query, and by Cauchy-Schwarz every score is off by at most
// eps_q = dE (1 + dq) + dq; margin_q = 2 eps_q + 1e-4 makes the survivor set a superset of the exact top-k
// (derivation: DESIGN.md section 3, K5).  Typical: margin 0.0046 instead of the worst-case 2^-6.
//
// The passes themselves run on the streaming kernel of msr_gemm_f32.hip (gemm_stream256_kernel<., BF16 = true>: rows from
// HBM / L2 straight into the register ring of the one wave that uses them, only the query image through LDS, up to four
// groups of 256 queries per launch co-scheduled per XCD).  This file holds what surrounds them: the image builders and
// margins, the tile-maximum transposition and k-th maximum, the bucket / candidate kernels.
//
// (The round-2 form of the pass -- 256 x 256 output tiles, rows AND queries through 160 KB of LDS by LDS-DMA, one barrier per
// K step: 0.44-0.46 of the bf16 peak, profiles/r02_gemm_knockout.md -- was removed in round 4; it is in the history at af12cb4.)
#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"
#include "msr_frag.h"
#include "msr_gemm_dev.h"

namespace {


// ---- image builders -----------------------------------------------------------------------------------------------
// dst[r] = bf16(src[r] * inv_norm[r]) for r < n_rows, zero rows up to n_pad (the GEMM reads 256 rows from a tile start).
// One wave per row at a time.  err_max (device word, bits of a non-negative float, atomicMax on the bits) receives
// max_r || bf16(u_r) - u_r ||_2 with u_r the normalised row: the measured rounding error of the image, from which the
// candidate margin is derived (msr_batch_margin) instead of the worst case 2^-8 ||u||.
__global__ __launch_bounds__(256) void unit_bf16_rows_kernel(const float* __restrict__ src, const float* __restrict__ inv_norm,
                                                              int64_t n_rows, int64_t n_pad, bf16x8* __restrict__ dst,
                                                              uint32_t* __restrict__ err_max) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    float worst = 0.f;
    for (int64_t r = wave; r < n_pad; r += n_waves) {
        float ss = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;                            // 16 B chunk of the bf16 row: 96 per row
            if (c >= MSR_DIM / 8) continue;
            bf16x8 v;
            if (r < n_rows) {
                const float sc = inv_norm ? inv_norm[r] : 1.0f;
                const f32x4 x = ((const f32x4*)(src + (size_t)r * MSR_DIM))[2 * c], y = ((const f32x4*)(src + (size_t)r * MSR_DIM))[2 * c + 1];
                const float u[8] = {x.x * sc, x.y * sc, x.z * sc, x.w * sc, y.x * sc, y.y * sc, y.z * sc, y.w * sc};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = (__bf16)u[j];
                    const float d = (float)v[j] - u[j];
                    ss += d * d;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
            }
            dst[(size_t)r * (MSR_DIM / 8) + c] = v;
        }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        worst = fmaxf(worst, ss);
    }
    if (lane == 0 && err_max) atomicMax(err_max, __float_as_uint(sqrtf(worst)));
}

// margin[q] = 2 eps_q + slack, eps_q = dE (1 + dq) + dq: |<e^, q^> - <e, q>| <= ||e^ - e|| ||q^|| + ||e|| ||q^ - q|| for
// unit e, q; dE = max row error of the image (err_max), dq = || bf16(q) - q || of THIS query, measured here.  slack covers
// the f32 accumulation of 768 products (<= 768 * 2^-24) and the rounding of the normalisations.
__global__ __launch_bounds__(256) void batch_margin_kernel(const float* __restrict__ qn, int nq, const uint32_t* __restrict__ err_max,
                                                            float* __restrict__ margin) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    float ss = 0.f;
    for (int j = lane; j < MSR_DIM; j += 64) {
        const float x = qn[(size_t)q * MSR_DIM + j];
        const float d = (float)(__bf16)x - x;
        ss += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) {
        const float dq = sqrtf(ss) * 1.0001f, dE = __uint_as_float(*err_max) * 1.0001f;
        margin[q] = 2.0f * (dE * (1.0f + dq) + dq * 1.000001f) + 1.0e-4f;
    }
}

// out[q][j] = max_p in[j][p][q]: the tile maxima as one row per query for the top-k select (32 x 32 LDS transpose)
__global__ __launch_bounds__(256) void gemm_tmax_kernel(const float* __restrict__ in, int n_j, int parts, int nq_pad,
                                                         float* __restrict__ out, int out_stride) {
    __shared__ float t[32][33];
    const int j0 = blockIdx.x * 32, q0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + ty + 8 * r;
        float v = -__builtin_inff();
        if (j < n_j)
            for (int p = 0; p < parts; ++p) v = fmaxf(v, in[((size_t)j * parts + p) * nq_pad + q0 + tx]);
        t[ty + 8 * r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + ty + 8 * r, j = j0 + tx;
        if (j < n_j) out[(size_t)q * out_stride + j] = t[tx][ty + 8 * r];
    }
}

// thr[q] = (k-th largest tile maximum) - margin; fewer than k finite maxima (or a padding query): +inf, i.e. no emission,
// and flag[q] = 1 so that the caller reports the query as "rerun on the exact path" (real queries only)
// thr[q] = (k-th largest valid value of row q of tmax) - margin[q] -- the job of a full top-k select + a threshold kernel
// (7 launches), done by ONE workgroup per query: a 3-pass radix select (11 + 11 + 10 bits of the orderable key) with an
// LDS histogram; the row (<= a few 10^4 tile maxima) is re-read from the L2 in every pass.  Fewer than k valid values:
// thr = +inf and flag[q] = 1 (the caller treats the query as overflowed).  Rows q >= nq (padding): +inf, flag 0.
__global__ __launch_bounds__(1024) void gemm_kth_kernel(const float* __restrict__ tmax, int n, int stride, int nq, int k,
                                                         const float* __restrict__ margin, float* __restrict__ thr,
                                                         int32_t* __restrict__ flag, float short_val, float margin_scale) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_bin, s_kk, s_hit;
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (q >= nq) {
        if (t == 0) { thr[q] = short_val; if (flag) flag[q] = 0; }
        return;
    }
    const float* row = tmax + (size_t)q * stride;
    uint32_t prefix = 0, mask = 0, kk = (uint32_t)k;
    bool short_row = false;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
        const uint32_t nb = pass == 2 ? 1024u : 2048u;
        hist[t] = 0; hist[t + 1024] = 0;
        if (t == 0) s_hit = 0;
        __syncthreads();
        for (int i = t; i < n; i += 1024) {
            const float x = row[i];
            if (!msr_valid(x)) continue;
            const uint32_t key = msr_ord32(x);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        // the bin that holds the kk-th largest key: thread t owns bins 2t (low) and 2t + 1 (high); counts above a thread =
        // higher lanes of its wave + higher waves
        const uint32_t h0 = 2 * t < (int)nb ? hist[2 * t] : 0u, h1 = 2 * t + 1 < (int)nb ? hist[2 * t + 1] : 0u;
        const uint32_t c2 = h0 + h1;
        uint32_t inc = c2;                                   // inclusive suffix sum over lanes >= lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_down(inc, o);
            if (lane + o < 64) inc += v;
        }
        if (lane == 0) wsum[wave] = inc;
        __syncthreads();
        uint32_t above = inc - c2;
        for (int w2 = wave + 1; w2 < 16; ++w2) above += wsum[w2];
        if (above < kk && kk <= above + c2) {                // exactly one thread, if the row holds kk candidates at all
            if (above + h1 >= kk) { s_bin = 2 * t + 1; s_kk = kk - above; }
            else { s_bin = 2 * t; s_kk = kk - above - h1; }
            s_hit = 1;
        }
        __syncthreads();
        if (!s_hit) { short_row = true; break; }             // (block-uniform)
        prefix |= s_bin << shift;
        mask |= (nb - 1) << shift;
        kk = s_kk;
        __syncthreads();
    }
    if (t == 0) {
        thr[q] = short_row ? short_val : msr_unord32(prefix) - (margin ? margin_scale * margin[q] : 0.0f);
        if (flag) flag[q] = short_row ? 1 : 0;
    }
}


// Workgroup buffers -> per-query lists of (row, score) with score >= thr2[q].
__global__ __launch_bounds__(256) void gemm_bucket_kernel(const int4* __restrict__ wgbuf, int wg_cap,
                                                           const int32_t* __restrict__ wg_count,   // (per wave)
                                                           const float* __restrict__ thr2, int2* __restrict__ pairs,
                                                           int pair_cap, int32_t* __restrict__ pair_n) {
    const int wg = blockIdx.y;
    int n = wg_count[wg];
    if (n > wg_cap) n = wg_cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int4 e = wgbuf[(size_t)wg * wg_cap + i];
        if (__int_as_float(e.z) >= thr2[e.y]) {
            const int pos = atomicAdd(&pair_n[e.y], 1);
            if (pos < pair_cap) pairs[(size_t)e.y * pair_cap + pos] = make_int2(e.x, e.z);
        }
    }
}

// One workgroup per query: (row, score) pairs -> per-document maxima -> documents within `margin` of the k-th largest
// maximum -> candidate list for the exact f32 rescoring (msr_batch.hip).  Overflow anywhere: cand_n = MSR_SEL_CAP + 1,
// which rescore_final_kernel reports as out_n = -1 (rerun on the exact path).
constexpr int GM_PAIR_CAP = 4096;
// Every candidate also gets the run of ITS emitted rows (cand_first_len[slot] = first | len << 13 into pairs[q][.].x, which
// is rewritten in (document, score) order): msr_batch_rescore_rows recomputes only those -- the arg-max row of a document
// whose exact max-cosine reaches the exact k-th score has an approximate score >= the bucket's bound, so it is among them
// (the argument in msr_gemm_f32.hip, gemm_f32_cand_kernel).  Both sorts carry a 32-bit payload for that.
static_assert(GM_PAIR_CAP <= (1 << 13), "cand_first_len packs first and len into 13 bits each");
__global__ __launch_bounds__(1024) void gemm_cand_kernel(int2* __restrict__ pairs, int32_t* __restrict__ pair_n,
                                                          const int32_t* __restrict__ chunk_doc, int64_t n_rows,
                                                          const int32_t* __restrict__ wg_count, int n_wg, int wg_cap,
                                                          const int32_t* __restrict__ flag, int k,
                                                          const float* __restrict__ margin,
                                                          int32_t* __restrict__ cand_doc, int32_t* __restrict__ cand_first_len,
                                                          int32_t* __restrict__ cand_n) {
    __shared__ uint64_t key[GM_PAIR_CAP];
    __shared__ uint32_t pay[GM_PAIR_CAP];               // sort 1: the entry's row; sort 2: the head's first | len << 13
    __shared__ int s_heads, s_over;
    const int q = blockIdx.x, t = threadIdx.x;
    const int raw = pair_n[q];
    // (all threads scan the per-wave counts: one thread walking 2048 words with a data-dependent exit took 0.12 ms)
    int over = raw > GM_PAIR_CAP || (flag && flag[q]);
    for (int i = t; i < n_wg; i += 1024) over |= wg_count[i] > wg_cap;
    if (t == 0) s_heads = 0;
    s_over = 0;
    __syncthreads();
    if (over) s_over = 1;
    __syncthreads();
    if (s_over) {
        if (t == 0) { cand_n[q] = MSR_SEL_CAP + 1; pair_n[q] = 0; }
        return;
    }
    int P = 64;
    while (P < raw) P <<= 1;
    auto sort_desc = [&]() {
        for (int kk = 2; kk <= P; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int idx = t; idx < (P >> 1); idx += 1024) {
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                    const int p = i | j;
                    const bool desc = (i & kk) == 0;
                    const uint64_t x = key[i], y = key[p];
                    if (desc ? x < y : x > y) {
                        key[i] = y; key[p] = x;
                        const uint32_t v = pay[i]; pay[i] = pay[p]; pay[p] = v;
                    }
                }
                __syncthreads();
            }
    };
    // (1) by (document, score) descending: the first entry of a document's run is its maximum
    for (int i = t; i < P; i += 1024) {
        uint64_t kk = 0;
        uint32_t row = 0;
        if (i < raw) {
            const int2 e = pairs[(size_t)q * GM_PAIR_CAP + i];
            if (e.x >= 0 && e.x < n_rows) {                 // (a row index is never trusted as an address)
                kk = ((uint64_t)(uint32_t)(chunk_doc[e.x] + 1) << 32) | msr_ord32(__int_as_float(e.y));   // doc + 1: 0 is the pad key
                row = (uint32_t)e.x;
            }
        }
        key[i] = kk; pay[i] = row;
    }
    __syncthreads();
    sort_desc();
    // (2) heads only, keyed by (score, ~document), with the run of the document's entries as payload; everything else
    // becomes the pad key.  The rows go back to pairs[q][.].x in the order of sort 1 (a document's rows adjacent).
    uint64_t mine[GM_PAIR_CAP / 1024];
    uint32_t mine_run[GM_PAIR_CAP / 1024];
#pragma unroll
    for (int u = 0; u < GM_PAIR_CAP / 1024; ++u) {
        const int i = t + u * 1024;
        uint64_t kk = 0;
        uint32_t run = 0;
        if (i < P && key[i] != 0) {
            pairs[(size_t)q * GM_PAIR_CAP + i].x = (int32_t)pay[i];
            const uint32_t d = (uint32_t)(key[i] >> 32);
            if (i == 0 || (uint32_t)(key[i - 1] >> 32) != d) {
                kk = ((uint64_t)(uint32_t)key[i] << 32) | (uint32_t)~(uint32_t)(d - 1);
                int len = 1;
                while (i + len < P && (uint32_t)(key[i + len] >> 32) == d) ++len;
                run = (uint32_t)i | ((uint32_t)len << 13);
            }
        }
        mine[u] = kk; mine_run[u] = run;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GM_PAIR_CAP / 1024; ++u) {
        const int i = t + u * 1024;
        if (i < P) {
            key[i] = mine[u]; pay[i] = mine_run[u];
            if (mine[u]) atomicAdd(&s_heads, 1);
        }
    }
    __syncthreads();
    sort_desc();
    const int heads = s_heads;
    // (3) the k-th largest per-document maximum is t (or -inf with fewer than k documents); keep >= t - margin
    const float tk = heads >= k ? msr_unord32((uint32_t)(key[k - 1] >> 32)) : -__builtin_inff();
    const float cut = tk - margin[q];
    __shared__ int s_keep;
    if (t == 0) s_keep = 0;
    __syncthreads();
    for (int i = t; i < heads; i += 1024)                                   // sorted: the kept ones are a prefix
        if (msr_unord32((uint32_t)(key[i] >> 32)) >= cut) {
            atomicAdd(&s_keep, 1);
            if (i < MSR_SEL_CAP) {
                cand_doc[(size_t)q * MSR_SEL_CAP + i] = (int32_t)~(uint32_t)key[i];
                cand_first_len[(size_t)q * MSR_SEL_CAP + i] = (int32_t)pay[i];
            }
        }
    __syncthreads();
    if (t == 0) { cand_n[q] = s_keep; pair_n[q] = 0; }
}

int g_gemm_dbg = 0;

}  // namespace

hipError_t msr_unit_bf16_rows(const float* src, const float* inv_norm, int64_t n_rows, int64_t n_pad, void* dst,
                              uint32_t* err_max, hipStream_t stream) {
    if (n_pad <= 0) return hipSuccess;
    hipError_t err = err_max ? hipMemsetAsync(err_max, 0, 4, stream) : hipSuccess;
    if (err != hipSuccess) return err;
    unit_bf16_rows_kernel<<<8192, 256, 0, stream>>>(src, inv_norm, n_rows, n_pad, (bf16x8*)dst, err_max);
    return hipGetLastError();
}

hipError_t msr_batch_margin(const float* qn, int nq, const uint32_t* err_max, float* margin, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    batch_margin_kernel<<<(nq + 3) / 4, 256, 0, stream>>>(qn, nq, err_max, margin);
    return hipGetLastError();
}

int msr_gemm_pair_cap() { return GM_PAIR_CAP; }

hipError_t msr_gemm_tmax(const float* tmax_t, int n_j, int parts, int nq_pad, float* out, int out_stride, hipStream_t stream) {
    if (n_j <= 0 || nq_pad <= 0) return hipSuccess;
    gemm_tmax_kernel<<<dim3((n_j + 31) / 32, nq_pad / 32), 256, 0, stream>>>(tmax_t, n_j, parts, nq_pad, out, out_stride);
    return hipGetLastError();
}
hipError_t msr_gemm_kth(const float* tmax, int n, int stride, int nq, int nq_pad, int k, const float* margin, float* thr,
                        int32_t* flag, hipStream_t stream, float short_val, float margin_scale) {
    if (nq_pad <= 0) return hipSuccess;
    gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>(tmax, n, stride, nq, k, margin, thr, flag, short_val, margin_scale);
    return hipGetLastError();
}
hipError_t msr_gemm_bucket(const void* wvbuf, int wv_cap, const int32_t* wv_count, int n_waves, const float* thr2,
                           void* pairs, int32_t* pair_n, hipStream_t stream) {
    gemm_bucket_kernel<<<dim3(2, (unsigned)n_waves), 256, 0, stream>>>((const int4*)wvbuf, wv_cap, wv_count, thr2, (int2*)pairs,
                                                                      GM_PAIR_CAP, pair_n);
    return hipGetLastError();
}
void msr_gemm_set_dbg(int v) { g_gemm_dbg = v; }

// The whole batched candidate path for nq <= g.max_queries queries (see the header of this file); ends with cand_doc /
// cand_first_len / cand_n filled for msr_batch_rescore_rows.  qn: normalised f32 queries [nq][768].
hipError_t msr_gemm_candidates(const GemmIndex& g, const DenseIndex& ix, const float* qn, int nq, int k, const float* margin,
                               int32_t* cand_doc, int32_t* cand_first_len, int32_t* cand_n, hipEvent_t* ev, hipStream_t stream) {
    const int nq_pad = (nq + 255) / 256 * 256;
    const int nt = nq_pad / 256;
    if (nq <= 0 || nq_pad > g.max_queries || k < 1 || g.n_tiles < 2 * k) return hipErrorInvalidValue;
    const int grid = g.n_cus / 8 * 8;
    if (grid < 8 || grid / 8 < nt) return hipErrorInvalidValue;
    hipError_t err;
    // ---- pass 1: every ss-th tile, tile maxima only ----
    int ss = g.n_tiles / (8 * k);
    ss = ss < 1 ? 1 : (ss > 16 ? 16 : ss);
#ifdef MSR_DIAG
    if (g_gemm_dbg & 256) ss = 1;                      // timing experiments: the sample pass covers every tile
#endif
    const int n_s = (g.n_tiles - ss / 2 + ss - 1) / ss;                   // tiles ss/2, ss/2 + ss, ...
    {
        // The pass: the 256-query streaming kernel of msr_gemm_f32.hip over the bf16 unit-row image, nt groups of 256
        // queries in ONE launch (rows through a register ring -- no LDS, no barrier for them --, the group's query image
        // through LDS; the nt workgroups that walk the same tiles share an XCD's L2).
        if ((err = msr_stream256_bf16_qimage(qn, nq, nt, g.qmat, stream)) != hipSuccess) return err;
        StreamArgs a{};
        a.E = (const char*)g.emb_n; a.inv_pad = nullptr; a.qimg = (const char*)g.qmat; a.tile_row = g.tile_row;
        a.n_rows = ix.n_chunks; a.tmax_t = g.tmax_t; a.nt = nt; a.q_base = 0; a.append = 0;
        a.t_first = ss / 2; a.t_stride = ss; a.t_count = n_s;
        if (ev && (err = hipEventRecord(ev[0], stream)) != hipSuccess) return err;
        if ((err = msr_stream256_bf16_launch(false, a, grid, stream)) != hipSuccess) return err;
        if (ev && (err = hipEventRecord(ev[1], stream)) != hipSuccess) return err;
        gemm_tmax_kernel<<<dim3((n_s + 31) / 32, nq_pad / 32), 256, 0, stream>>>(g.tmax_t, n_s, 1, nq_pad, (float*)g.tmax, g.tmax_stride);
        gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>((const float*)g.tmax, n_s, g.tmax_stride, nq, k, margin, g.thr, g.flag, __builtin_inff(), 1.0f);
        a.t_first = 0; a.t_stride = 1; a.t_count = g.n_tiles;
        a.thr = g.thr; a.wvbuf = g.wgbuf; a.wv_cap = g.wv_cap; a.wv_count = g.wv_count;
        if (ev && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
        if ((err = msr_stream256_bf16_launch(true, a, grid, stream)) != hipSuccess) return err;
        if (ev && (err = hipEventRecord(ev[3], stream)) != hipSuccess) return err;
        gemm_tmax_kernel<<<dim3((g.n_tiles + 31) / 32, nq_pad / 32), 256, 0, stream>>>(g.tmax_t, g.n_tiles, 1, nq_pad, (float*)g.tmax, g.tmax_stride);
        gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>((const float*)g.tmax, g.n_tiles, g.tmax_stride, nq, k, margin, g.thr2, nullptr, __builtin_inff(), 1.0f);
    }
    // ---- finish: bucket, per-document maxima, candidates ----
    gemm_bucket_kernel<<<dim3(2, (unsigned)grid * 8), 256, 0, stream>>>((const int4*)g.wgbuf, g.wv_cap, g.wv_count, g.thr2,
                                                                       (int2*)g.pairs, GM_PAIR_CAP, g.pair_n);
    gemm_cand_kernel<<<nq, 1024, 0, stream>>>((int2*)g.pairs, g.pair_n, ix.chunk_doc, ix.n_chunks, g.wv_count, grid * 8, g.wv_cap,
                                              g.flag, k, margin, cand_doc, cand_first_len, cand_n);
    return hipGetLastError();
}
