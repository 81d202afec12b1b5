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
// -DMSR_DIAG builds only: the round-2 kernel (256 x 256 output tiles, rows AND queries through 160 KB of LDS by LDS-DMA,
// one barrier per K step; 0.44-0.46 of the bf16 peak) stays as the A/B partner of the streaming pass (msr_tune(100, 512)).
#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"
#include "msr_frag.h"
#include "msr_gemm_dev.h"

namespace {

#ifdef MSR_DIAG
constexpr int GM_THREADS = 512;
constexpr int GM_KT = MSR_DIM / 64;            // K steps per output tile
constexpr int GM_HALF = 16384;                 // bytes of one half-tile image: 128 rows x 128 B
constexpr int GM_LDS3 = 10 * GM_HALF;           // 3 x (A-lo, A-hi) + 2 x (B-lo, B-hi) = all 160 KB
constexpr int GM_ROWB = MSR_DIM * 2;           // bytes per bf16 row

struct GemmArgs {
    const char* A;             // bf16 [rows + pad][768], unit rows
    const char* B;             // bf16 [nq_pad][768], unit rows (zero rows as padding)
    const int32_t* tile_row;   // [n_tiles + 1]
    int t_first, t_stride, t_count;   // tiles of this pass: t_first + j t_stride, j < t_count
    int nt;                    // query tiles (nq_pad / 256)
    float* tmax_t;             // [t_count][2 wave rows][nq_pad] maxima of the tile's rows owned by wave row 0 / 1 (every cell
                               // is written exactly once per pass, 64 B per store; gemm_tmax_kernel transposes and joins them)
    int nq_pad;
    const float* thr;          // [nq_pad] emit threshold (+inf: never)                       -- emit pass only
    int4* wgbuf;               // [gridDim.x * 8 waves][wv_cap] {row, query, score bits, tile}  -- emit pass only
    int wv_cap;
    int32_t* wv_count;         // [gridDim.x * 8] entries each wave produced (may exceed wv_cap: overflow)
    int dbg;                   // -DMSR_DIAG builds only (timing experiments, results are wrong): bit 0 = every tile reads
                               // the rows of tile 0 (A always from cache), bit 1 = B always K step 0
};

template <bool EMIT>
__global__ __launch_bounds__(GM_THREADS) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;                 // the wave's 128 x 64 piece of the 256 x 256 tile: rows wr, queries wc
    const int li16 = lane & 15, lg = lane >> 4;

    // ---- which tiles: the nt workgroups of a row-tile group share blockIdx % 8 (one XCD under round-robin dispatch) ----
    const int per_x = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int gpx = per_x / a.nt;
    const int mg = li / a.nt, nt = li - mg * a.nt;
    int wave_cnt = 0;                                  // entries this wave has emitted (wave-uniform)
    int4* wvbuf = EMIT ? a.wgbuf + ((size_t)blockIdx.x * 8 + w) * a.wv_cap : nullptr;
    const int G = 8 * gpx, gid = xcd * gpx + mg;
    const bool active = mg < gpx && gid < a.t_count;
    if (!active) {                                     // workgroup-uniform
        if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = 0;
        return;
    }
    const int n_mine = (a.t_count - gid + G - 1) / G;

    // ---- per-lane constants ----
    uint32_t goff[2];                                  // DMA source offsets inside a 128-row half-tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (2 * w + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (4 * i + (lane >> 4));       // (R >> 1) & 7 == 4 i + (lane >> 4)
        goff[i] = (uint32_t)(R * GM_ROWB + c * 16);
    }
    uint32_t foff[2];                                  // fragment read offsets inside a 16-row block
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = (uint32_t)(li16 * 128 + (((ks * 4 + lg) ^ ((lane >> 1) & 7)) << 4));
    const uint32_t a_base = (uint32_t)(wr * 64 * 128), b_base = (uint32_t)(wc * 32 * 128);

#ifdef MSR_DIAG
    const int dbg = a.dbg;                             // bit 2: no DMA after the prologue, bit 3: no MFMA, bit 4: no fragment reads
#else
    constexpr int dbg = 0;
#endif
    bool prologue = true;
    auto stage = [&](const char* src, int slot) {      // one half-tile: 2 x 1 KiB per wave
        if ((dbg & 4) && !prologue) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + goff[i]), (lds_void*)(smem + slot + (2 * w + i) * 1024), 16, 0, 0);
    };

    float thrv[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            thrv[nh][ni] = EMIT ? a.thr[nt * 256 + nh * 128 + wc * 32 + ni * 16 + li16] : 0.f;

    // (make the compiler wait for these ordinary loads HERE: a pending VGPR load next to in-flight LDS-DMAs would make it
    // drain the whole DMA pipeline at the first use, in every epilogue)
    asm volatile("" :: "v"(thrv[0][0]), "v"(thrv[0][1]), "v"(thrv[1][0]), "v"(thrv[1][1]));

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mh][mi][nh][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_of = [&](int j) { return a.t_first + j * a.t_stride; };
#ifdef MSR_DIAG
    auto first_row = [&](int j) { return (a.dbg & 1) ? 0 : a.tile_row[tile_of(j)]; };
#else
    auto first_row = [&](int j) { return a.tile_row[tile_of(j)]; };
#endif
    int jt = gid;
    int row0 = first_row(jt), row_end = a.tile_row[tile_of(jt) + 1];
    int jn = jt + G < a.t_count ? jt + G : jt;
    int row0n = first_row(jn);
    const char* Bq = a.B + (size_t)(nt * 256) * GM_ROWB;

    // ---- prologue: the first K step of the first tile and the row halves of the second ----
    // LDS slots: row buffer j (0..2) at j * 32 KB (lo, hi), query buffer d (0..1) at 96 KB + d * 32 KB (lo, hi)
    auto a3 = [](int j, int hi) { return (2 * j + hi) * GM_HALF; };
    auto b3 = [](int d, int hi) { return (6 + 2 * d + hi) * GM_HALF; };
    stage(a.A + (size_t)row0 * GM_ROWB, a3(0, 0));
    stage(Bq, b3(0, 0));
    stage(a.A + (size_t)(row0 + 128) * GM_ROWB, a3(0, 1));
    stage(Bq + 128 * GM_ROWB, b3(0, 1));
    stage(a.A + (size_t)row0 * GM_ROWB + 128, a3(1, 0));
    stage(a.A + (size_t)(row0 + 128) * GM_ROWB + 128, a3(1, 1));
    wait_vm0();
    wg_barrier();
    prologue = false;

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ======== the K loop: ONE barrier per K step, rows three buffers deep ========
    // History (profiles/r02_gemm_knockout.md): a first version ran one 64 x 32 quadrant per phase with 8 barriers per
    // K step; with MFMAs, fragment reads and DMAs all removed it still took 41 % of its time -- a workgroup barrier
    // costs ~200 cycles during which no wave issues anything.  Now a K step is one uninterrupted stretch of 64 MFMAs
    // per wave: inside it a wave alternates fragment reads and 16-MFMA groups -- (row half, k half) sub-steps: 4 A +
    // 4 B fragments, the next A fragments read into a second register set before the current MFMAs -- and the two
    // waves of a SIMD drift apart freely, so one wave's LDS latency hides behind the other's MFMAs.  12 barriers per tile.
    bf16x8 a0[4] = {}, a1[4] = {}, b4[4] = {};
    auto mma2 = [&](const bf16x8 (&aa)[4], auto mh_c, auto zero_c) {
        constexpr int mh = decltype(mh_c)::value;
        constexpr bool ZERO = decltype(zero_c)::value != 0;     // first touch of these accumulators in a tile: C = 0
        if (dbg & 8) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mh][mi][nh][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        aa[mi], b4[nh * 2 + ni], ZERO ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mh][mi][nh][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // With only the DMAs left, a K step with two buffers took 1.8 us -- the 64 KB of a step do not arrive within the
    // step's 64 MFMAs; 78 % of the reads hit the XCD's L2 (the queries always, the rows for three of the four workgroups
    // that share a row tile), but each step waits for its slowest line, and the row tile of the leading workgroup comes
    // from HBM.  So the rows get two steps of flight: step u issues the queries of step u + 1 FIRST and the rows of step
    // u + 2 after them; the wait before the barrier is vmcnt(4), which retires everything but those 4 youngest DMAs --
    // the in-order counter then never makes the rows wait for the queries.
    // LDS: 3 x 32 KB of rows + 2 x 32 KB of queries = all 160 KB.
    auto rd_a3 = [&](bf16x8 (&dst)[4], int j, int mh, int ks) {
        if (dbg & 16) return;
        const char* p = smem + a3(j, mh) + a_base + foff[ks];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) dst[mi] = *(const bf16x8*)(p + mi * 2048);
    };
    auto rd_b3 = [&](int d, int ks) {
        if (dbg & 16) return;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                b4[nh * 2 + ni] = *(const bf16x8*)(smem + b3(d, nh) + b_base + ni * 2048 + foff[ks]);
    };
    auto kstep3 = [&](auto j_c, auto d_c, auto first_c, const char* An2, const char* Bn1) {
        constexpr int j = decltype(j_c)::value, d = decltype(d_c)::value;
        stage(Bn1, b3(d ^ 1, 0));
        stage(Bn1 + 128 * GM_ROWB, b3(d ^ 1, 1));
        stage(An2, a3((j + 2) % 3, 0));
        stage(An2 + 128 * GM_ROWB, a3((j + 2) % 3, 1));
        rd_b3(d, 0);
        rd_a3(a0, j, 0, 0);
        rd_a3(a1, j, 1, 0);
        mma2(a0, I0{}, first_c);                       // the first K step of a tile starts its sums from 0: no clearing pass
        rd_a3(a0, j, 0, 1);
        mma2(a1, I1{}, first_c);
        rd_b3(d, 1);
        rd_a3(a1, j, 1, 1);
        mma2(a0, I0{}, I0{});
        mma2(a1, I1{}, I0{});
        wait_vm4();
        wg_barrier();
    };

    const float NEG_INF = -__builtin_inff();
    for (int it = 0; it < n_mine; ++it) {
        const char* A0 = a.A + (size_t)row0 * GM_ROWB;
        const char* A1 = a.A + (size_t)row0n * GM_ROWB;
        {
            using I2 = std::integral_constant<int, 2>;
            // K step kt of this tile (kt = 6 k6 + s): rows of step kt + 2 (maybe of the next tile), queries of step kt + 1
            auto a_src = [&](int kt2) { return kt2 < GM_KT ? A0 + kt2 * 128 : A1 + (kt2 - GM_KT) * 128; };
            auto b_src = [&](int kt1) { return Bq + (kt1 < GM_KT ? kt1 : kt1 - GM_KT) * 128; };
#pragma unroll 1
            for (int k6 = 0; k6 < GM_KT / 6; ++k6) {     // (all 12 steps spelled out cost 90 spilled registers)
                const int kt = 6 * k6;
                kstep3(I0{}, I0{}, I0{}, a_src(kt + 2), b_src(kt + 1));
                kstep3(I1{}, I1{}, I0{}, a_src(kt + 3), b_src(kt + 2));
                kstep3(I2{}, I0{}, I0{}, a_src(kt + 4), b_src(kt + 3));
                kstep3(I0{}, I1{}, I0{}, a_src(kt + 5), b_src(kt + 4));
                kstep3(I1{}, I0{}, I0{}, a_src(kt + 6), b_src(kt + 5));
                kstep3(I2{}, I1{}, I0{}, a_src(kt + 7), b_src(kt + 6));
            }
        }
        // ---- epilogue: accumulator (mh, mi, nh, ni)[rr] = row mh 128 + wr 64 + mi 16 + 4 lg + rr of the tile,
        //      query nt 256 + nh 128 + wc 32 + ni 16 + li16 ----
        int n_valid = row_end - row0;                                      // rows of THIS tile (the rest belongs to the next)
#ifdef MSR_DIAG
        if (a.dbg & 1) n_valid = 250;
#endif
        // (opaque copy of the lane's column index: keeps the compiler from hoisting the epilogue's address arithmetic
        // out of the tile loop, where it would be spilled -- and a scratch reload waits for vmcnt(0), DMAs included)
        int col_e = li16;
        asm volatile("" : "+v"(col_e));
        float cmax[2][2] = {{NEG_INF, NEG_INF}, {NEG_INF, NEG_INF}};
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int blk = mh * 128 + wr * 64 + mi * 16;
                if (blk >= n_valid || (dbg & 32)) continue;                // wave-uniform
                const int rb = blk + 4 * lg;
                const bool part = blk + 16 > n_valid;                      // wave-uniform
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        f32x4 v = acc[mh][mi][nh][ni];
                        if (part) {                                        // (at most one block per wave and tile)
                            asm volatile("" ::: "memory");                 // a real branch: do not predicate this into every block
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr)
                                if (rb + rr >= n_valid) v[rr] = NEG_INF;
                        }
                        const float m = max3_raw(max2_raw(v[0], v[1]), v[2], v[3]);
                        cmax[nh][ni] = max2_raw(cmax[nh][ni], m);
                        // Emission: every branch below is WAVE-UNIFORM (ballots), the position comes from a per-wave counter
                        // kept in a scalar register and a prefix count over the emitting lanes -- no LDS or global atomic
                        // (the compiler orders an LDS atomic behind ALL pending LDS-DMAs: s_waitcnt vmcnt(0))
                        if (EMIT && __ballot(m >= thrv[nh][ni]) != 0) {
                            const int q = nt * 256 + nh * 128 + wc * 32 + ni * 16 + col_e;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const bool hit = v[rr] >= thrv[nh][ni];
                                const unsigned long long hm = __ballot(hit);
                                if (hm != 0) {
                                    const int pos = wave_cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                                    if (hit && pos < a.wv_cap)
                                        wvbuf[pos] = make_int4(row0 + rb + rr, q, __float_as_int(v[rr]), tile_of(jt));
                                    wave_cnt += __popcll(hm);
                                }
                            }
                        }
                    }
            }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                // max over the 4 lanes that hold the same query (lane, lane ^ 16, lane ^ 32, lane ^ 48): two lane-swap
                // instructions, no LDS round trip
                float m = cmax[nh][ni];
                auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = max2_raw(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
                auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = max2_raw(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
                if (lg == 0 && !(dbg & 64))
                    a.tmax_t[((size_t)jt * 2 + wr) * a.nq_pad + nt * 256 + nh * 128 + wc * 32 + ni * 16 + col_e] = m;
            }
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mh][mi][nh][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // next tile
        jt = jn;
        row0 = row0n;
        row_end = a.tile_row[tile_of(jt) + 1];
        jn = jt + G < a.t_count ? jt + G : jt;
        row0n = first_row(jn);
    }
    wait_vm0();                                        // the DMAs issued for a step that never runs
    if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = wave_cnt;
}

#endif  // MSR_DIAG (the round-2 kernel)

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
                                                         int32_t* __restrict__ flag) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_bin, s_kk, s_hit;
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (q >= nq) {
        if (t == 0) { thr[q] = __builtin_inff(); if (flag) flag[q] = 0; }
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
        thr[q] = short_row ? __builtin_inff() : msr_unord32(prefix) - (margin ? margin[q] : 0.0f);
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
__global__ __launch_bounds__(1024) void gemm_cand_kernel(const int2* __restrict__ pairs, int32_t* __restrict__ pair_n,
                                                          const int32_t* __restrict__ chunk_doc,
                                                          const int32_t* __restrict__ wg_count, int n_wg, int wg_cap,
                                                          const int32_t* __restrict__ flag, int k,
                                                          const float* __restrict__ margin,
                                                          int32_t* __restrict__ cand_doc, int32_t* __restrict__ cand_n) {
    __shared__ uint64_t key[GM_PAIR_CAP];
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
                    if (desc ? x < y : x > y) { key[i] = y; key[p] = x; }
                }
                __syncthreads();
            }
    };
    // (1) by (document, score) descending: the first entry of a document's run is its maximum
    for (int i = t; i < P; i += 1024) {
        uint64_t kk = 0;
        if (i < raw) {
            const int2 e = pairs[(size_t)q * GM_PAIR_CAP + i];
            kk = ((uint64_t)(uint32_t)(chunk_doc[e.x] + 1) << 32) | msr_ord32(__int_as_float(e.y));   // doc + 1: 0 is the pad key
        }
        key[i] = kk;
    }
    __syncthreads();
    sort_desc();
    // (2) heads only, keyed by (score, ~document); everything else becomes the pad key
    uint64_t mine[GM_PAIR_CAP / 1024];
#pragma unroll
    for (int u = 0; u < GM_PAIR_CAP / 1024; ++u) {
        const int i = t + u * 1024;
        uint64_t kk = 0;
        if (i < P && key[i] != 0 && (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)))
            kk = ((uint64_t)(uint32_t)key[i] << 32) | (uint32_t)~(uint32_t)((key[i] >> 32) - 1);
        mine[u] = kk;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GM_PAIR_CAP / 1024; ++u) {
        const int i = t + u * 1024;
        if (i < P) {
            key[i] = mine[u];
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
            if (i < MSR_SEL_CAP) cand_doc[(size_t)q * MSR_SEL_CAP + i] = (int32_t)~(uint32_t)key[i];
        }
    __syncthreads();
    if (t == 0) { cand_n[q] = s_keep; pair_n[q] = 0; }
}

#ifdef MSR_DIAG
// qmat[q] = bf16(qn[q]) for q < nq, zero rows up to nq_pad
__global__ __launch_bounds__(256) void qmat_kernel(const float* __restrict__ qn, int nq, int nq_pad, bf16x8* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq_pad * (MSR_DIM / 8)) return;
    const int q = i / (MSR_DIM / 8);
    bf16x8 v;
    if (q < nq) {
        const f32x4 x = ((const f32x4*)qn)[2 * (size_t)i], y = ((const f32x4*)qn)[2 * (size_t)i + 1];
        v[0] = (__bf16)x.x; v[1] = (__bf16)x.y; v[2] = (__bf16)x.z; v[3] = (__bf16)x.w;
        v[4] = (__bf16)y.x; v[5] = (__bf16)y.y; v[6] = (__bf16)y.z; v[7] = (__bf16)y.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
    }
    dst[i] = v;
}

#endif
int g_gemm_dbg = 0;

#ifdef MSR_DIAG
template <bool EMIT>
hipError_t launch_gemm_t(const GemmArgs& a, int grid, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t err = hipFuncSetAttribute((const void*)gemm_kernel<EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             GM_LDS3);
        if (err != hipSuccess) return err;
        attr_done = true;
    }
    GemmArgs b = a;
    b.dbg = g_gemm_dbg;
    gemm_kernel<EMIT><<<grid, GM_THREADS, GM_LDS3, stream>>>(b);
    return hipGetLastError();
}

hipError_t launch_gemm(bool emit, const GemmArgs& a, int grid, hipStream_t stream) {
    return emit ? launch_gemm_t<true>(a, grid, stream) : launch_gemm_t<false>(a, grid, stream);
}

#endif

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
                        int32_t* flag, hipStream_t stream) {
    if (nq_pad <= 0) return hipSuccess;
    gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>(tmax, n, stride, nq, k, margin, thr, flag);
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
// cand_n filled for msr_batch_rescore.  qn: normalised f32 queries [nq][768].
hipError_t msr_gemm_candidates(const GemmIndex& g, const DenseIndex& ix, const float* qn, int nq, int k, const float* margin,
                               int32_t* cand_doc, int32_t* cand_n, hipEvent_t* ev, hipStream_t stream) {
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
#ifdef MSR_DIAG
    if (g_gemm_dbg & 512) {                             // timing experiments: the 256 x 256 LDS-tiled kernel of round 2
        qmat_kernel<<<(nq_pad * (MSR_DIM / 8) + 255) / 256, 256, 0, stream>>>(qn, nq, nq_pad, (bf16x8*)g.qmat);
        GemmArgs a{};
        a.A = (const char*)g.emb_n; a.B = (const char*)g.qmat; a.tile_row = g.tile_row; a.nt = nt;
        a.tmax_t = g.tmax_t; a.nq_pad = nq_pad;
        a.t_first = ss / 2; a.t_stride = ss; a.t_count = n_s;
        if (ev && (err = hipEventRecord(ev[0], stream)) != hipSuccess) return err;
        if ((err = launch_gemm(false, a, grid, stream)) != hipSuccess) return err;
        if (ev && (err = hipEventRecord(ev[1], stream)) != hipSuccess) return err;
        gemm_tmax_kernel<<<dim3((n_s + 31) / 32, nq_pad / 32), 256, 0, stream>>>(g.tmax_t, n_s, 2, nq_pad, (float*)g.tmax, g.tmax_stride);
        gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>((const float*)g.tmax, n_s, g.tmax_stride, nq, k, margin, g.thr, g.flag);
        a.t_first = 0; a.t_stride = 1; a.t_count = g.n_tiles;
        a.thr = g.thr; a.wgbuf = (int4*)g.wgbuf; a.wv_cap = g.wv_cap; a.wv_count = g.wv_count;
        if (ev && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
        if ((err = launch_gemm(true, a, grid, stream)) != hipSuccess) return err;
        if (ev && (err = hipEventRecord(ev[3], stream)) != hipSuccess) return err;
        gemm_tmax_kernel<<<dim3((g.n_tiles + 31) / 32, nq_pad / 32), 256, 0, stream>>>(g.tmax_t, g.n_tiles, 2, nq_pad, (float*)g.tmax, g.tmax_stride);
        gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>((const float*)g.tmax, g.n_tiles, g.tmax_stride, nq, k, margin, g.thr2, nullptr);
    } else
#endif
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
        gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>((const float*)g.tmax, n_s, g.tmax_stride, nq, k, margin, g.thr, g.flag);
        a.t_first = 0; a.t_stride = 1; a.t_count = g.n_tiles;
        a.thr = g.thr; a.wvbuf = g.wgbuf; a.wv_cap = g.wv_cap; a.wv_count = g.wv_count;
        if (ev && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
        if ((err = msr_stream256_bf16_launch(true, a, grid, stream)) != hipSuccess) return err;
        if (ev && (err = hipEventRecord(ev[3], stream)) != hipSuccess) return err;
        gemm_tmax_kernel<<<dim3((g.n_tiles + 31) / 32, nq_pad / 32), 256, 0, stream>>>(g.tmax_t, g.n_tiles, 1, nq_pad, (float*)g.tmax, g.tmax_stride);
        gemm_kth_kernel<<<nq_pad, 1024, 0, stream>>>((const float*)g.tmax, g.n_tiles, g.tmax_stride, nq, k, margin, g.thr2, nullptr);
    }
    // ---- finish: bucket, per-document maxima, candidates ----
    gemm_bucket_kernel<<<dim3(2, (unsigned)grid * 8), 256, 0, stream>>>((const int4*)g.wgbuf, g.wv_cap, g.wv_count, g.thr2,
                                                                       (int2*)g.pairs, GM_PAIR_CAP, g.pair_n);
    gemm_cand_kernel<<<nq, 1024, 0, stream>>>((const int2*)g.pairs, g.pair_n, ix.chunk_doc, g.wv_count, grid * 8, g.wv_cap,
                                              g.flag, k, margin, cand_doc, cand_n);
    return hipGetLastError();
}
