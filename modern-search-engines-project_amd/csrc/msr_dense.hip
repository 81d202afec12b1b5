// K2 + K3 -- query x chunk cosine over the whole embedding matrix with a fused per-document max-pool.
//
// Replaces the reference's cosine_similarity calls (reranker/reranker_api.py:273-287) and per-document
// arg-max (:370) at full-corpus scale, i.e. the missing Retriever.quick_search (search_api.py:60,87).
//
// Shape of the work: E is [C][768] f32 (15.36 GB at C = 5 M) and is read exactly once per launch for up to
// 32 queries -> HBM-bound (arithmetic intensity Q/2 flop/B).  The dot products run on the exact-f32 matrix
// cores: v_mfma_f32_16x16x4_f32 with 16 chunk rows as the A operand and 16 queries as the B operand, so
// each lane needs ONE f32 of E per MFMA and a 16 B/lane global load feeds four MFMAs.  The k index of an
// MFMA is only a label: lane (i = lane & 15, g = lane >> 4) loads E[row i][16t + 4g .. +3] and the query
// image in LDS is stored in the same permuted order, so no data ever moves between lanes.
//
// dense_scan_v2_kernel ("wave streaming", described at its definition): every wave streams its own span of rows,
// no workgroup barriers; also instantiated for bf16 rows (batched path, <= 32 queries or a per-document row limit).
// Spans are cut at document boundaries with equal chunk counts, so documents never cross spans and no
// inter-workgroup communication exists.  The default for calls without a row limit is the K-split kernel of
// msr_dense_ks.hip; this kernel serves what that one does not take (row limit, interleaved layout, corpora that
// fail its ring precondition).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"
#include "msr_frag.h"

namespace {

constexpr int KSTEPS = MSR_DIM / 16;                // 48 float4 per lane per row group

// ---------------------------------------------------------------------------------------------------------
// Scan kernel ("wave streaming"): every WAVE streams its own span of row groups with no workgroup
// barrier after the query image is loaded.  Per 16-row group: 64 B of every row per k-step as one 16 B/lane
// load (register double buffer that runs ACROSS groups, so the next group's first batch is in flight during
// this group's epilogue), the MFMAs, then the 16 x Q cosines go through a wave-private LDS tile so that lane
// q holds query q's column and walks the 16 rows; document boundaries are wave-uniform scalars (v_readlane of
// chunk_doc), the running maximum of the open document simply stays in a register across groups.  Finished
// documents are staged in a wave-private LDS buffer and written 32 documents at a time, 128 B per query row.
//
// The same kernel serves two element types:
//   f32  : 16 dims per k-step, 4 x v_mfma_f32_16x16x4_f32 per load and query block, up to 32 queries, exact f32
//   bf16 : 32 dims per k-step, 1 x v_mfma_f32_16x16x32_bf16 per load and query block, up to 64 queries; used as
//          the candidate generator of the batched path (K5), its scores are re-computed in f32 afterwards

// MODE: how a row group is multiplied
//   0  f32 rows, v_mfma_f32_16x16x4_f32 (exact f32: bit-for-bit a k-ordered fmaf chain)
//   1  bf16 rows, v_mfma_f32_16x16x32_bf16 (candidate generator of the batched path)
//   2  f32 rows split on the fly into two f16 pieces (x = hi + lo, |x - hi - lo| <= 2^-20 |x|); three
//      v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi) replace eight f32 MFMAs.  Every f16 x f16 product is exact in
//      f32 and the accumulation is f32, so |error| <= 3 * 2^-20 * sum|e_i q_i| <= 2.9e-6 for unit vectors: inside
//      the 1e-5 cosine tolerance by construction, at about a third of the matrix-pipe time and far less power.

template <int QB, int MODE, int WAVES, int OBD = 32> struct ScanCfgV2 {   // OBD: staged documents per wave (0 = none)
    static constexpr int NL = MODE == MODE_BF16 ? MSR_DIM / 32 : MSR_DIM / 16;      // 16 B loads per lane per row group
    static constexpr int KS = MODE == MODE_F32 ? MSR_DIM / 16 : MSR_DIM / 32;       // MFMA k-steps per row group
    static constexpr int QPIECES = MODE == MODE_F16X2 ? 2 : 1;                      // operand pieces per k-step
    static constexpr int ROW16 = MODE == MODE_BF16 ? MSR_DIM * 2 / 16 : MSR_DIM * 4 / 16;   // 16-byte units per row
    static constexpr int NQP = 16 * QB;
    static constexpr int SROW = NQP + 1;
    static constexpr int THREADS = WAVES * 64;
    static constexpr size_t q_bytes = (size_t)QB * KS * QPIECES * 64 * 16;
    static constexpr size_t t_bytes = (size_t)16 * SROW * 4;                 // per wave: 16 rows x queries
    static constexpr size_t o_bytes = (size_t)OBD * SROW * 4;                // per wave: staged documents
    static constexpr size_t wave_bytes = (t_bytes + o_bytes + 15) & ~(size_t)15;
    static constexpr size_t total = q_bytes + WAVES * wave_bytes;
};

// qimg: the query image already in fragment order, [QB][KS][64 lanes] x 16 B (see build_qimage_kernel)
template <int QB, bool TILED, int LB, int MODE, int WAVES, int OBD = 32, int G = 1>
__global__ __launch_bounds__(WAVES * 64) void dense_scan_v2_kernel(DenseIndex ix, const void* __restrict__ emb,
                                                                    const int32_t* __restrict__ wspan,
                                                                    int n_wspans, const f32x4* __restrict__ qimg,
                                                                    int nq, int max_chunks,
                                                                    float* __restrict__ docscore, int dbg) {
    using L = ScanCfgV2<QB, MODE, WAVES, OBD>;
    constexpr int KS = L::KS, NL = L::NL;
    constexpr bool BF16 = MODE == MODE_BF16;
    constexpr bool DIRECT = OBD == 0;
    static_assert(OBD == 0 || OBD == 8 || OBD == 16 || OBD == 32, "staging depth");
    static_assert(NL % LB == 0 && (MODE != MODE_F16X2 || LB % 2 == 0), "whole load batches per group");
    static_assert(!(BF16 && TILED), "the interleaved image exists for f32 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Qs = (f32x4*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* T = (float*)(smem + L::q_bytes + (size_t)w * L::wave_bytes);      // [16][SROW]
    float* OB = T + 16 * L::SROW;                                            // [OBD][SROW]
    const int li = lane & 15, lg = lane >> 4;

    for (int idx = tid; idx < QB * KS * L::QPIECES * 64; idx += L::THREADS) Qs[idx] = qimg[idx];
    __syncthreads();                                             // the only workgroup barrier

    const int s = blockIdx.x * WAVES + w;
    if (s >= n_wspans) return;                                   // wave-uniform
    if (ix.gate && *ix.gate == 0) return;                        // a fallback launch that is not needed (msr_engine.hip)
    const int64_t C = ix.n_chunks;
    const float NEG_INF = -__builtin_inff();
    const int d0 = wspan[s], d1 = wspan[s + 1];
    const int64_t c0 = ix.doc_off[d0], c1 = ix.doc_off[d1];

    // staged output: documents ob_base .. ob_base + ob_n - 1 (consecutive) for every query
    int ob_base = d0, ob_n = 0;
    auto flush = [&]() {
        // lane -> (query sub-index, document): 64 lanes cover (64 / OBD) queries x OBD documents per instruction
        constexpr int OD = OBD > 0 ? OBD : 32;
        constexpr int QPI = 64 / OD;
        for (int qq0 = 0; qq0 < nq; qq0 += QPI) {
            const int qq = qq0 + lane / OD, dd = lane % OD;
            if (qq < nq && dd < ob_n && !(dbg & 1)) docscore[(int64_t)qq * ix.score_stride + ob_base + dd] = OB[dd * L::SROW + qq];
        }
        ob_base += ob_n;
        ob_n = 0;
    };
    auto emit = [&](float m) {                                   // lane q holds the value of query q
        if (DIRECT) {
            // 64-query sweeps have no LDS left for staging: each lane stores its query's value; the 32 stores that
            // complete a 128 B line of a score row come from this same wave within ~10 row groups (L2 merges them)
            if (lane < nq) docscore[(int64_t)lane * ix.score_stride + ob_base] = m;
            ++ob_base;
            return;
        }
        if (lane < L::NQP) OB[ob_n * L::SROW + lane] = m;
        // flush when the staged run reaches a 128 B boundary of the score rows: full aligned lines, no partial-sector
        // writes (they cost a read-modify-write at the ECC memory: 12 % of the 32-query sweep in profile r01_h)
        constexpr int OD = OBD > 0 ? OBD : 32;
        ++ob_n;
        if (((ob_base + ob_n) & (OD - 1)) == 0) flush();
    };

    int cur_doc = d0 - 1;                                        // last document that has been emitted/opened
    bool open = false;
    int cnt = 0;                                                 // rows of the open document seen so far
    float m = NEG_INF;

    if (c1 > c0) {
        constexpr int PSTRIDE = TILED ? 64 : 4;                  // 16-byte units between k-steps
        constexpr int NBATCH = NL / LB;
        const int64_t g0 = c0 >> 4, g1 = (c1 + 15) >> 4;
        auto row_ptr = [&](int64_t grp) -> const f32x4* {
            if (TILED) return (const f32x4*)emb + (size_t)grp * (16 * L::ROW16) + lane;
            int64_t r = grp * 16 + li;
            if (r > C - 1) r = C - 1;
            return (const f32x4*)emb + (size_t)r * L::ROW16 + lg;
        };
        auto meta_row = [&](int64_t grp) -> int64_t {            // row whose chunk_doc / inv_norm this lane fetches
            int64_t r = grp * 16 + li;
            return r > C - 1 ? C - 1 : r;
        };
        // A wave works on a UNIT of G consecutive row groups at a time: every query fragment read from LDS is used
        // for G x 16 rows, which divides the LDS read traffic (the cost that grows with the number of queries) by G.
        auto clampg = [&](int64_t g) { return g < g1 ? g : g1 - 1; };     // a unit may stick out of the span: masked later
        f32x4 buf0[G][LB], buf1[G][LB];
        const f32x4* p[G];
        int dv[G];
        float iv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            p[g] = row_ptr(clampg(g0 + g));
#pragma unroll
            for (int u = 0; u < LB; ++u) buf0[g][u] = p[g][(size_t)u * PSTRIDE];
            dv[g] = ix.chunk_doc[meta_row(clampg(g0 + g))];
            iv[g] = ix.inv_norm[meta_row(clampg(g0 + g))];
        }
        __builtin_amdgcn_sched_barrier(0);
        // One unit.  Batch b of the unit (b = 0 .. NBATCH-1) lives in buf[(b + PH) & 1]; PH is the parity the unit
        // starts with.  With an even NBATCH it is always 0; with an odd one (LB == NL: the whole next unit is
        // prefetched while this one is consumed) the loop below alternates PH = 0, 1 so that every register array is
        // indexed statically.
        auto body = [&](auto ph_c, int64_t grp) {
            constexpr int PH = decltype(ph_c)::value;
            const f32x4* pn[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pn[g] = row_ptr(clampg(grp + G + g));
            const bool has_next = grp + G < g1;
            // Independent accumulation chains: back-to-back MFMAs on ONE accumulator stall on its 40-cycle dependent
            // latency, so consecutive MFMAs alternate between chains (query blocks x groups); a lone chain (one query
            // block, one group) is split in two that are added at the end.
            constexpr int NCH = (QB == 1 && G == 1) ? 2 : QB;
            f32x4 acc[G][NCH];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int dv_next[G];
            float iv_next[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { dv_next[g] = dv[g]; iv_next[g] = iv[g]; }
#pragma unroll
            for (int nb = 0; nb < NBATCH; ++nb) {
                const bool into1 = ((nb + 1 + PH) & 1) != 0;     // buffer of the batch being prefetched
                if (nb + 1 < NBATCH) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int u = 0; u < LB; ++u) {
                            const f32x4 x = p[g][(size_t)((nb + 1) * LB + u) * PSTRIDE];
                            if (into1) buf1[g][u] = x; else buf0[g][u] = x;
                        }
                } else if (has_next) {                           // first batch of the next unit
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int u = 0; u < LB; ++u) {
                            const f32x4 x = pn[g][(size_t)u * PSTRIDE];
                            if (into1) buf1[g][u] = x; else buf0[g][u] = x;
                        }
                        dv_next[g] = ix.chunk_doc[meta_row(clampg(grp + G + g))];
                        iv_next[g] = ix.inv_norm[meta_row(clampg(grp + G + g))];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == MODE_F16X2) {
#pragma unroll
                    for (int u = 0; u < LB; u += 2) {
                        const int t = (nb * LB + u) >> 1;
                        f16x8 ahi[G], alo[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const f32x4 a0 = ((nb + PH) & 1) ? buf1[g][u] : buf0[g][u];
                            const f32x4 a1 = ((nb + PH) & 1) ? buf1[g][u + 1] : buf0[g][u + 1];
                            split_f16(a0, a1, ahi[g], alo[g]);
                        }
                        f16x8 bhi[QB], blo[QB];
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            bhi[qb] = __builtin_bit_cast(f16x8, Qs[((qb * KS + t) * 2 + 0) * 64 + lane]);
                            blo[qb] = __builtin_bit_cast(f16x8, Qs[((qb * KS + t) * 2 + 1) * 64 + lane]);
                        }
                        if constexpr (NCH > QB) {                // one block, one group: two chains
                            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[0], bhi[0], acc[0][0], 0, 0, 0);
                            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[0], blo[0], acc[0][1], 0, 0, 0);
                            acc[0][t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[0], bhi[0], acc[0][t & 1], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int piece = 0; piece < 3; ++piece)
#pragma unroll
                                for (int g = 0; g < G; ++g)
#pragma unroll
                                    for (int qb = 0; qb < QB; ++qb)
                                        acc[g][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                            piece == 0 ? alo[g] : ahi[g], piece == 1 ? blo[qb] : bhi[qb], acc[g][qb], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < LB; ++u) {
                        const int t = nb * LB + u;
                        f32x4 a[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) a[g] = ((nb + PH) & 1) ? buf1[g][u] : buf0[g][u];
                        f32x4 bq[QB];
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) bq[qb] = Qs[(qb * KS + t) * 64 + lane];
                        if constexpr (BF16) {
#pragma unroll
                            for (int g = 0; g < G; ++g)
#pragma unroll
                                for (int qb = 0; qb < QB; ++qb) {
                                    const int ch = NCH > QB ? (t & 1) : qb;
                                    acc[g][ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, a[g]), __builtin_bit_cast(bf16x8, bq[qb]), acc[g][ch], 0, 0, 0);
                                }
                        } else if constexpr (NCH > QB) {
                            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].x, bq[0].x, acc[0][0], 0, 0, 0);
                            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].y, bq[0].y, acc[0][1], 0, 0, 0);
                            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].z, bq[0].z, acc[0][0], 0, 0, 0);
                            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].w, bq[0].w, acc[0][1], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
#pragma unroll
                                for (int g = 0; g < G; ++g)
#pragma unroll
                                    for (int qb = 0; qb < QB; ++qb)
                                        acc[g][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][c], bq[qb][c], acc[g][qb], 0, 0, 0);
                        }
                    }
                }
            }
            if constexpr (NCH > QB) acc[0][0] = acc[0][0] + acc[0][1];       // join the two chains
            // ---- epilogue of the unit's groups (wave-private; no barrier: a wave's LDS ops execute in order) ----
            // D layout: lane (li = query column, lg) holds rows 4 lg + reg.  inv_norm of row r sits in lane r.
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float inv = __shfl(iv[g], 4 * lg + reg);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) T[(4 * lg + reg) * L::SROW + 16 * qb + li] = acc[g][qb][reg] * inv;
                }
                const int64_t row0 = (grp + g) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = row0 + r;
                    const int d = __builtin_amdgcn_readlane(dv[g], r);   // wave-uniform
                    if (row < c0 || row >= c1) continue;         // rows of a neighbouring span / past the span
                    if (!open || d != cur_doc) {
                        if (open) emit(m);
                        for (int e = cur_doc + 1; e < d; ++e) emit(NEG_INF);     // chunk-less documents in between
                        cur_doc = d; open = true; cnt = 0; m = NEG_INF;
                    }
                    if (max_chunks == 0 || cnt < max_chunks) {
                        const float v = lane < L::NQP ? T[r * L::SROW + lane] : NEG_INF;
                        m = fmaxf(m, v);
                    }
                    ++cnt;
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) { p[g] = pn[g]; dv[g] = dv_next[g]; iv[g] = iv_next[g]; }
        };
        for (int64_t grp = g0; grp < g1;) {
            body(std::integral_constant<int, 0>{}, grp);
            grp += G;
            if constexpr ((NBATCH & 1) != 0) {
                if (grp >= g1) break;
                body(std::integral_constant<int, 1>{}, grp);
                grp += G;
            }
        }
    }
    if (open) emit(m);
    for (int e = cur_doc + 1; e < d1; ++e) emit(NEG_INF);        // trailing chunk-less documents
    if (ob_n) flush();
}

// Query image in MFMA-fragment order.  f32: block (qb, t) lane l = qn[16 qb + (l & 15)][16 t + 4 (l >> 4) .. +3];
// bf16: lane l = bf16(qn[16 qb + (l & 15)][32 t + 8 (l >> 4) .. +7]) (round-to-nearest-even).
template <int MODE>
__global__ __launch_bounds__(256) void build_qimage_kernel(const float* __restrict__ qn, int n_blocks,
                                                            f32x4* __restrict__ qimg) {
    constexpr int KS = MODE == MODE_F32 ? MSR_DIM / 16 : MSR_DIM / 32;
    const int idx = blockIdx.x * 256 + threadIdx.x;              // one (block, k-step, lane)
    if (idx >= n_blocks * KS * 64) return;
    const int l = idx & 63;
    const int t = (idx >> 6) % KS;
    const int qb = idx / (KS * 64);
    const float* src = qn + (size_t)(16 * qb + (l & 15)) * MSR_DIM;
    if (MODE == MODE_BF16) {
        const float* s8 = src + 32 * t + 8 * (l >> 4);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)s8[j];
        qimg[idx] = __builtin_bit_cast(f32x4, v);
    } else if (MODE == MODE_F16X2) {
        // the 8 k-elements of lane (i, g) at k-step t are what the row side loads as two float4:
        // dims 16 (2t) + 4g .. +3 and 16 (2t + 1) + 4g .. +3
        const f32x4 a = *(const f32x4*)(src + 16 * (2 * t) + 4 * (l >> 4));
        const f32x4 b = *(const f32x4*)(src + 16 * (2 * t + 1) + 4 * (l >> 4));
        f16x8 hi, lo;
        split_f16(a, b, hi, lo);
        qimg[(size_t)(idx >> 6) * 128 + l] = __builtin_bit_cast(f32x4, hi);        // [qb][t][piece][lane]
        qimg[(size_t)(idx >> 6) * 128 + 64 + l] = __builtin_bit_cast(f32x4, lo);
    } else {
        qimg[idx] = *(const f32x4*)(src + 16 * t + 4 * (l >> 4));
    }
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ dst, int64_t n, float value) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = value;
}

__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, int nq,
                                                            float* __restrict__ qn, int nq_pad) {
    // one wave per (padded) query row: qn = q / ||q||, zero norm -> divide by 1 (sklearn normalize)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nq_pad) return;
    float v[12];
    float ss = 0.f;
    for (int j = 0; j < 12; ++j) {
        v[j] = row < nq ? q[(size_t)row * MSR_DIM + lane + 64 * j] : 0.f;
        ss += v[j] * v[j];
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    float nrm = sqrtf(ss);
    if (nrm == 0.f) nrm = 1.f;
    for (int j = 0; j < 12; ++j) qn[(size_t)row * MSR_DIM + lane + 64 * j] = v[j] / nrm;
}

__global__ void fill_chunk_doc_kernel(const int32_t* __restrict__ doc_off, int64_t n_docs,
                                      int32_t* __restrict__ chunk_doc) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    for (int64_t c = doc_off[d]; c < doc_off[d + 1]; ++c) chunk_doc[c] = (int32_t)d;
}

__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ emb, int64_t n_rows,
                                                            float* __restrict__ inv_norm) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const f32x4* p = (const f32x4*)(emb + (size_t)row * MSR_DIM);
    float ss = 0.f;
    for (int j = 0; j < 3; ++j) {
        const f32x4 v = p[lane + 64 * j];
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    float nrm = sqrtf(ss);
    if (nrm == 0.f) nrm = 1.f;
    if (lane == 0) inv_norm[row] = 1.0f / nrm;
}

// dst block (group, t) is 64 lanes x float4: lane l = 16 g + i holds src[16 group + i][16 t + 4 g .. +3]
__global__ __launch_bounds__(256) void interleave_kernel(const float* __restrict__ src, int64_t n_rows,
                                                          float* __restrict__ dst, int64_t n_vec) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vec;
         v += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(v & 63);
        const int64_t bt = v >> 6;
        const int t = (int)(bt % KSTEPS);
        const int64_t grp = bt / KSTEPS;
        const int64_t r = grp * 16 + (l & 15);
        f32x4 x = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r < n_rows) x = *(const f32x4*)(src + (size_t)r * MSR_DIM + 16 * t + 4 * (l >> 4));
        ((f32x4*)dst)[v] = x;
    }
}

// One wave per (query, winner): recompute the winner's chunk cosines and report the first maximum.
template <bool TILED>
__global__ __launch_bounds__(64) void best_chunk_kernel(DenseIndex ix, const float* __restrict__ qn, int k,
                                                         int max_chunks, const int32_t* __restrict__ out_doc,
                                                         const int32_t* __restrict__ out_n,
                                                         int32_t* __restrict__ out_chunk, int per_wave) {
    // (per_wave > 1: the gated fallback launch, which almost never has work -- a sixteenth of the waves to start and retire)
    const int q = blockIdx.y, lane = threadIdx.x;
    if (ix.gate && ix.gate[ix.gate_per64 ? q >> 6 : 0] == 0) return;
    float qv[12];
    for (int j = 0; j < 12; ++j) qv[j] = qn[(size_t)q * MSR_DIM + lane + 64 * j];
    const int r_end = (int)(blockIdx.x + 1) * per_wave < k ? (int)(blockIdx.x + 1) * per_wave : k;
    for (int r = blockIdx.x * per_wave; r < r_end; ++r) {
    if (r >= out_n[q]) {
        if (lane == 0) out_chunk[(int64_t)q * k + r] = -1;
        continue;
    }
    const int d = out_doc[(int64_t)q * k + r];
    const int64_t ds = ix.doc_off[d];
    int64_t de = ix.doc_off[d + 1];
    if (max_chunks > 0 && ds + max_chunks < de) de = ds + max_chunks;
    float best = -__builtin_inff();
    int64_t arg = -1;
    for (int64_t c = ds; c < de; ++c) {
        float s = 0.f;
        for (int j = 0; j < 12; ++j) {
            const int dim = lane + 64 * j;
            size_t off;
            if (TILED) {
                const int t = dim >> 4, g = (dim >> 2) & 3, e = dim & 3;
                off = (size_t)(c >> 4) * (16 * MSR_DIM) + ((size_t)t * 64 + g * 16 + (c & 15)) * 4 + e;
            } else {
                off = (size_t)c * MSR_DIM + dim;
            }
            s += ix.emb[off] * qv[j];
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        s *= ix.inv_norm[c];
        if (s > best) { best = s; arg = c; }
    }
    if (lane == 0) out_chunk[(int64_t)q * k + r] = (int32_t)arg;
    }
}

// Diagnostic build only (-DMSR_DIAG, tools/ab_scan.py): MSR_SCAN_DEBUG bit 0 drops the score-row stores for timing
// experiments.  The product library reads no environment variable.
static int scan_debug_flags() {
#ifdef MSR_DIAG
    static const int v = [] { const char* e = getenv("MSR_SCAN_DEBUG"); return e ? atoi(e) : 0; }();
    return v;
#else
    return 0;
#endif
}

template <int QB, bool TILED, int LB, int WAVES = 8, int OBD = 32, int MODE = MODE_F32, int G = 1>
hipError_t launch_scan_v2(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                          hipStream_t stream) {
    using L = ScanCfgV2<QB, MODE, WAVES, OBD>;
    static_assert(L::total <= 160 * 1024, "LDS budget");
    const size_t lds = L::total;
    hipError_t err = hipFuncSetAttribute((const void*)dense_scan_v2_kernel<QB, TILED, LB, MODE, WAVES, OBD, G>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    const int n_img = QB * L::KS * 64;
    build_qimage_kernel<MODE><<<(n_img + 255) / 256, 256, 0, stream>>>(qn, QB, (f32x4*)ix.qimg);
    // per-wave span tables exist for 8 and for 12 waves per CU
    const int32_t* spans = WAVES == 12 ? ix.wspan12_doc : ix.wspan_doc;
    const int n_sp = WAVES == 12 ? ix.n_wspans12 : ix.n_wspans;
    const int grid = (n_sp + WAVES - 1) / WAVES;
    dense_scan_v2_kernel<QB, TILED, LB, MODE, WAVES, OBD, G><<<grid, L::THREADS, lds, stream>>>(
        ix, ix.emb, spans, n_sp, (const f32x4*)ix.qimg, nq, max_chunks, docscore, scan_debug_flags());
    return hipGetLastError();
}

template <int QB, int WAVES, int OBD, int LB = 12>
hipError_t launch_scan_bf16_cfg(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                                hipStream_t stream) {
    using L = ScanCfgV2<QB, MODE_BF16, WAVES, OBD>;
    const size_t lds = L::total;
    static_assert(L::total <= 160 * 1024, "LDS budget");
    hipError_t err = hipFuncSetAttribute((const void*)dense_scan_v2_kernel<QB, false, LB, MODE_BF16, WAVES, OBD>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    const int n_img = QB * L::KS * 64;
    build_qimage_kernel<MODE_BF16><<<(n_img + 255) / 256, 256, 0, stream>>>(qn, QB, (f32x4*)ix.qimg);
    // the per-wave spans were cut for 8 waves per CU; with fewer waves a workgroup takes fewer of them
    const int grid = (ix.n_wspans + WAVES - 1) / WAVES;
    dense_scan_v2_kernel<QB, false, LB, MODE_BF16, WAVES, OBD><<<grid, L::THREADS, lds, stream>>>(
        ix, ix.emb_bf16, ix.wspan_doc, ix.n_wspans, (const f32x4*)ix.qimg, nq, max_chunks, docscore, scan_debug_flags());
    return hipGetLastError();
}

template <int QB>
hipError_t launch_scan_bf16(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                            hipStream_t stream) {
    if constexpr (QB <= 2) return launch_scan_bf16_cfg<QB, 8, 32>(ix, qn, nq, max_chunks, docscore, stream);
    // 48/64 queries: the query image takes 72/96 KB of LDS, which leaves room for 4 waves with 32 staged documents
#ifdef MSR_DIAG
    // (A/B knob of the diagnostic build: 8 waves with 8 staged documents / direct stores / whole-group prefetch)
    static const int knob = [] { const char* v = getenv("MSR_BF16_CFG"); return v ? atoi(v) : 0; }();
    if (knob == 1) return launch_scan_bf16_cfg<QB, 8, 8>(ix, qn, nq, max_chunks, docscore, stream);
    if (knob == 2) return launch_scan_bf16_cfg<QB, 8, 0>(ix, qn, nq, max_chunks, docscore, stream);
    if (knob == 3) return launch_scan_bf16_cfg<QB, 4, 32, 24>(ix, qn, nq, max_chunks, docscore, stream);
#endif
    return launch_scan_bf16_cfg<QB, 4, 32>(ix, qn, nq, max_chunks, docscore, stream);
}

template <int QB, bool TILED>
hipError_t dispatch_variant(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                            hipStream_t stream) {
    // the engine resolves scan_variant 0 at bind time to 7 / 14 (f16 split) or 2 (exact f32), see msr_bind_chunks
    if (ix.variant == 2) return launch_scan_v2<QB, TILED, 12>(ix, qn, nq, max_chunks, docscore, stream);   // exact f32 MFMA
    return launch_scan_v2<QB, TILED, 12, 8, 32, MODE_F16X2>(ix, qn, nq, max_chunks, docscore, stream);     // f16 split
}

// min / max of inv_norm over all rows (positive floats order like their bit patterns)
__global__ __launch_bounds__(256) void norm_range_kernel(const float* __restrict__ inv_norm, int64_t n,
                                                          uint32_t* __restrict__ out) {
    uint32_t lo = 0x7F800000u, hi = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t u = __float_as_uint(inv_norm[i]);
        lo = u < lo ? u : lo; hi = u > hi ? u : hi;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

}  // namespace

hipError_t msr_dense_scan(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                          hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    // variants 14 / 15: K-split kernel (msr_dense_ks.hip), f16-split products
    if ((ix.variant == 14 || ix.variant == 15) && ix.wide_ok && ix.layout == 0 && max_chunks == 0)
        return msr_dense_scan_wide(ix, qn, nq, docscore, stream);
    // exact f32 products: the K-split kernel pays off above 32 queries (matrix-core bound: 4.5 ms per 64 queries against
    // 2 x 2.8 ms on the narrow kernel; at <= 32 queries the narrow kernel is faster)
    if (ix.variant == 2 && nq > 32 && ix.wide_ok && ix.layout == 0 && max_chunks == 0)
        return msr_dense_scan_wide_exact(ix, qn, nq, docscore, stream);
    if (nq > 32) return hipErrorInvalidValue;
    const bool tiled = ix.layout == 1;
    if (nq <= 16)
        return tiled ? dispatch_variant<1, true>(ix, qn, nq, max_chunks, docscore, stream)
                     : dispatch_variant<1, false>(ix, qn, nq, max_chunks, docscore, stream);
    return tiled ? dispatch_variant<2, true>(ix, qn, nq, max_chunks, docscore, stream)
                 : dispatch_variant<2, false>(ix, qn, nq, max_chunks, docscore, stream);
}

hipError_t msr_build_qimage(int mode, const float* qn, int n_blocks, void* qimg, hipStream_t stream) {
    const int ks = mode == MODE_F32 ? MSR_DIM / 16 : MSR_DIM / 32;
    const int n_img = n_blocks * ks * 64;
    const int grid = (n_img + 255) / 256;
    if (mode == MODE_F32) build_qimage_kernel<MODE_F32><<<grid, 256, 0, stream>>>(qn, n_blocks, (f32x4*)qimg);
    else if (mode == MODE_BF16) build_qimage_kernel<MODE_BF16><<<grid, 256, 0, stream>>>(qn, n_blocks, (f32x4*)qimg);
    else build_qimage_kernel<MODE_F16X2><<<grid, 256, 0, stream>>>(qn, n_blocks, (f32x4*)qimg);
    return hipGetLastError();
}

hipError_t msr_inv_norm_range(const float* inv_norm, int64_t n, uint32_t* out2, hipStream_t stream) {
    const uint32_t init[2] = {0x7F800000u, 0u};
    hipError_t err = hipMemcpyAsync(out2, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (err != hipSuccess) return err;
    if (n > 0) norm_range_kernel<<<1024, 256, 0, stream>>>(inv_norm, n, out2);
    return hipGetLastError();
}

hipError_t msr_dense_scan_bf16(const DenseIndex& ix, const float* qn, int nq, int max_chunks, float* docscore,
                               hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    if (nq > 64 || !ix.emb_bf16) return hipErrorInvalidValue;
    if (nq <= 16) return launch_scan_bf16<1>(ix, qn, nq, max_chunks, docscore, stream);
    if (nq <= 32) return launch_scan_bf16<2>(ix, qn, nq, max_chunks, docscore, stream);
    if (nq <= 48) return launch_scan_bf16<3>(ix, qn, nq, max_chunks, docscore, stream);
    return launch_scan_bf16<4>(ix, qn, nq, max_chunks, docscore, stream);
}

hipError_t msr_fill_f32(float* dst, int64_t n, float value, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    fill_f32_kernel<<<(unsigned)std::min<int64_t>((n + 255) / 256, 65536), 256, 0, stream>>>(dst, n, value);
    return hipGetLastError();
}

hipError_t msr_prep_queries(const float* q, int nq, float* qn, int nq_pad, hipStream_t stream) {
    if (nq_pad <= 0) return hipSuccess;
    prep_queries_kernel<<<(nq_pad + 3) / 4, 256, 0, stream>>>(q, nq, qn, nq_pad);
    return hipGetLastError();
}

hipError_t msr_fill_chunk_doc(const int32_t* doc_off, int64_t n_docs, int32_t* chunk_doc, hipStream_t stream) {
    if (n_docs <= 0) return hipSuccess;
    fill_chunk_doc_kernel<<<(unsigned)((n_docs + 255) / 256), 256, 0, stream>>>(doc_off, n_docs, chunk_doc);
    return hipGetLastError();
}

hipError_t msr_row_inv_norm(const float* emb, int64_t n_rows, float* inv_norm, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    row_inv_norm_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, stream>>>(emb, n_rows, inv_norm);
    return hipGetLastError();
}

hipError_t msr_interleave(const float* src, int64_t n_rows, float* dst, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    const int64_t n_vec = ((n_rows + 15) / 16) * KSTEPS * 64;
    int64_t blocks = (n_vec + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    interleave_kernel<<<(unsigned)blocks, 256, 0, stream>>>(src, n_rows, dst, n_vec);
    return hipGetLastError();
}

hipError_t msr_best_chunk(const DenseIndex& ix, const float* qn, int nq, int k, int max_chunks,
                          const int32_t* out_doc, const int32_t* out_n, int32_t* out_chunk, hipStream_t stream) {
    if (nq <= 0 || k <= 0) return hipSuccess;
    const int per_wave = ix.gate ? 16 : 1;
    dim3 grid((unsigned)((k + per_wave - 1) / per_wave), (unsigned)nq);
    if (ix.layout == 1)
        best_chunk_kernel<true><<<grid, 64, 0, stream>>>(ix, qn, k, max_chunks, out_doc, out_n, out_chunk, per_wave);
    else
        best_chunk_kernel<false><<<grid, 64, 0, stream>>>(ix, qn, k, max_chunks, out_doc, out_n, out_chunk, per_wave);
    return hipGetLastError();
}
