// K2 + K3 + K4 for batches of 65 .. 128 queries: the dense scan as ONE streaming pass over the ORIGINAL f32 rows (gfx950).
//
// Same cosine and per-document max as the sweeps (reference: reranker/reranker_api.py:285, :370; Retriever.quick_search,
// search_api.py:60,87), for 128 queries per pass over E (the sweep: 64) -- a 128-query step reads the 15.36 GB once.
//
// Arithmetic: a FILTER pass with one f16 product per element, then EXACT f32 cosines for the few documents that can
// matter.  The pass multiplies f16(e) (round to nearest, converted in registers; E stays f32 in HBM, no second image)
// by f16(q^), f32 accumulation on the matrix cores: v_mfma_f32_16x16x32_f16, 2 x 768 flop per (row, query) -- a third of
// the f16x2-split arithmetic of the sweeps, which made this pass matrix-bound.  Its scores s^ are approximate with a
// MEASURED bound: dE = max_r ||e_r - f16(e_r)|| / ||e_r|| (one pass at bind time, f16_row_error_kernel), dq = the same
// for the normalised query (f16_margin_kernel); |s^ - s| <= eps_q = dE (1 + dq) + dq (Cauchy-Schwarz twice, the argument
// of msr_batch.hip).  With t^ the k-th largest per-document s^, every document of the exact top-k has
// s^ >= t^ - 2 eps_q, so keeping everything above (a lower bound of t^) - margin_q, margin_q = 2 eps_q + 1e-4, loses
// nothing; the survivors' cosines are then recomputed in plain f32 from the f32 rows (msr_batch_rescore_rows: the emitted
// rows of each candidate document) and sorted exactly.  Typical margin: 1.1e-3.  Every RETURNED score is an exact f32 cosine.
//
// No score matrix, no score rows: row tiles (<= 256 rows) are cut at document boundaries, so the k-th largest TILE
// MAXIMUM is attained by k different documents -- a lower bound of t^.  Pass 1 (every ss-th tile) computes tile maxima
// only and gives the emission threshold; pass 2 (all tiles) stores the maxima of ALL tiles (a tighter bound afterwards)
// and appends every (row, query, score) at or above the threshold to a buffer private to the WAVE (scalar counter +
// lane prefix count, plain 16 B stores: no atomics).  Finish: entries above the tighter bound are bucketed per query
// (msr_gemm_bucket), reduced to distinct documents (gemm_f32_cand_kernel), rescored and sorted.  A query whose entries do
// not fit anywhere on the way (huge tie groups) raises a device-side gate; the engine's sweeps, always enqueued behind
// this path, run only when that gate is up (no host round trip).
//
// Kernel shape (gemm_stream_kernel; one persistent workgroup per CU, 8 waves, tile 256 rows x 128 queries, K step 32):
// wave w owns rows 32 w .. +32 of the tile and ALL 128 queries, so it is the ONLY reader of its rows -- they never pass
// through LDS: every wave loads its rows straight into a register ring, 4 K steps (16 KB per wave, 128 KB per CU) in
// flight.  LDS holds only what the waves share, the f16 query image: blocks of 8 K steps (64 KB), double buffered, filled
// by LDS-DMA one block ahead.  The workgroup meets at ONE barrier per block (3 per tile); in between the waves drift
// freely, so one wave's memory wait hides behind the others' MFMAs.  (The first version staged the rows through LDS with
// a barrier per K step: 3.6 ms per pass with the split arithmetic, 3.1 ms with one product; this form: 2.7 ms =
// 5.7 TB/s on the same box, profiles/r02_gemm_knockout.md.)
#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"
#include "msr_frag.h"
#include "msr_gemm_dev.h"

namespace {

constexpr int GF_THREADS = 512;
constexpr int GF_KT = MSR_DIM / 32;             // 24 K steps per tile
constexpr int GF_ROWB = MSR_DIM * 4;            // bytes per f32 row
constexpr int GF_PAIR_CAP = 4096;

typedef StreamArgs GemmF32Args;          // (msr_internal.h: shared with the bf16 candidate pass of msr_gemm.hip)

// a 64-bit value the compiler can prove wave-uniform (scalar registers): lets the DMA use the saddr + 32-bit voffset form
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

constexpr int GS_NB = 8;                        // K steps per query block
constexpr int GS_STEP = 8192;                   // bytes of one K step of the f16 query image: 128 queries x 64 B
constexpr int GS_BLK = GS_NB * GS_STEP;         // 64 KB
constexpr int GS_LDS = 2 * GS_BLK;              // 128 KB
constexpr int GS_D = 4;                         // K steps of rows in flight per wave

typedef const __attribute__((address_space(1))) char* gptr;       // (global, not flat: flat loads also count as LDS ops)
// a 16-byte global load the compiler neither moves nor counts; the caller waits (s_waitcnt vmcnt) and pins before use
template <int OFF>
__device__ __forceinline__ void gload16(f32x4& r, gptr p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pin4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void pin2(f32x4& a, f32x4& b) { asm volatile("" : "+v"(a), "+v"(b)); }

template <bool EMIT>
__global__ __launch_bounds__(GF_THREADS) void gemm_stream_kernel(GemmF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave w owns rows 32 w .. 32 w + 32 of the tile, ALL 128 queries
    const int li16 = lane & 15, lg = lane >> 4;
    const int G = (int)gridDim.x, gid = (int)blockIdx.x;
    int wave_cnt = EMIT && a.append ? __builtin_amdgcn_readfirstlane(a.wv_count[blockIdx.x * 8 + w]) : 0;
    int4* wvbuf = EMIT ? (int4*)a.wvbuf + ((size_t)blockIdx.x * 8 + w) * a.wv_cap : nullptr;
    if (gid >= a.t_count) {                             // workgroup-uniform
        if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = wave_cnt;
        return;
    }
    const int n_mine = (a.t_count - gid + G - 1) / G;
#ifdef MSR_DIAG
    const int dbg = a.dbg;
#else
    constexpr int dbg = 0;
#endif
    // query fragments: query 16 ni + li16, logical 16 B chunk lg of its 64 B, stored at physical chunk lg ^ (((q >> 3) & 1) << 1)
    const uint32_t foffB = (uint32_t)(li16 * 64 + ((lg ^ (((li16 >> 3) & 1) << 1)) << 4));
    // one query block: 64 KB, linear; wave w moves 8 KB of it
    auto stage_b = [&](int blk, int pb) {
        const char* base = uniform_ptr(a.qimg + (size_t)blk * GS_BLK + (size_t)w * 8192);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (uint32_t)(i * 1024 + lane * 16)),
                                             (lds_void*)(smem + pb * GS_BLK + w * 8192 + i * 1024), 16, 0, 0);
    };

    float thrv[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) thrv[ni] = EMIT ? a.thr[ni * 16 + li16] : 0.f;

    f32x4 acc[2][8];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_of = [&](int j) { return a.t_first + j * a.t_stride; };
    // the lane's two rows of a tile (fragment mi: row 32 w + 16 mi + li16; rows past the end of the matrix re-read the last
    // row, they are masked in the epilogue), at its 32-byte piece lg of a K step
    auto row_ptr = [&](int row0, int mi) -> gptr {
        int64_t r = (int64_t)row0 + 32 * w + 16 * mi + li16;
        if (r > a.n_rows - 1) r = a.n_rows - 1;
        return (gptr)a.E + (size_t)r * GF_ROWB + lg * 32;
    };
    int jt = gid;
    int row0 = a.tile_row[tile_of(jt)], row_end = a.tile_row[tile_of(jt) + 1];
    int jn = jt + G < a.t_count ? jt + G : jt;
    int row0n = a.tile_row[tile_of(jn)];
    gptr rp0 = row_ptr(row0, 0), rp1 = row_ptr(row0, 1);
    gptr rn0 = row_ptr(row0n, 0), rn1 = row_ptr(row0n, 1);
    const __attribute__((address_space(1))) float* invp = (const __attribute__((address_space(1))) float*)a.inv_pad + w * 32 + 4 * lg;

    // The row ring is driven by hand: the loads are inline asm (the compiler neither reorders them nor counts them) and
    // every use is preceded by an explicit s_waitcnt.  vmcnt retires in order, so "all but the N youngest" is exact:
    // per wave and query block the order is [8 DMAs of the next block] then per step [wait, convert, 4 row loads, compute].
    // The rows of step s8 were loaded four steps earlier; younger than them are the loads of three steps (12) and, for the
    // first four steps of a block, that block's 8 DMAs as well (20).  (The two inverse-norm loads of a tile's last block
    // are not counted: leaving fewer in flight than strictly possible is always safe.)
    f32x4 ring[GS_D][2][2];                              // [slot][fragment][16-byte half]
    auto load_rows = [&](auto slot_c, gptr p0, gptr p1, auto off_c) {
        constexpr int slot = decltype(slot_c)::value, off = decltype(off_c)::value;
        if (dbg & 4) return;
        gload16<off>(ring[slot][0][0], p0);
        gload16<off + 16>(ring[slot][0][1], p0);
        gload16<off>(ring[slot][1][0], p1);
        gload16<off + 16>(ring[slot][1][1], p1);
    };
    auto pin_rows = [&](auto slot_c) {                   // after a wait: the slot's registers now hold the loaded rows
        constexpr int slot = decltype(slot_c)::value;
        pin4(ring[slot][0][0], ring[slot][0][1], ring[slot][1][0], ring[slot][1][1]);
    };
    // ---- prologue: query block 0, rows of steps 0 .. GS_D - 1 ----
    stage_b(0, 0);
#pragma unroll
    for (int s = 0; s < GS_D; ++s)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) ring[s][mi][0] = ring[s][mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_rows(std::integral_constant<int, 0>{}, rp0, rp1, std::integral_constant<int, 0>{});
    load_rows(std::integral_constant<int, 1>{}, rp0, rp1, std::integral_constant<int, 128>{});
    load_rows(std::integral_constant<int, 2>{}, rp0, rp1, std::integral_constant<int, 256>{});
    load_rows(std::integral_constant<int, 3>{}, rp0, rp1, std::integral_constant<int, 384>{});
    wait_vm0();
    pin_rows(std::integral_constant<int, 0>{}); pin_rows(std::integral_constant<int, 1>{});
    pin_rows(std::integral_constant<int, 2>{}); pin_rows(std::integral_constant<int, 3>{});
    wg_barrier();
    int pb = 0;                                          // LDS buffer of the current query block

    f32x4 inv4[2];
    inv4[0] = inv4[1] = (f32x4){1.f, 1.f, 1.f, 1.f};
    const float NEG_INF = -__builtin_inff();
    // one K step; S8: position in the query block (compile time), the ring slot is S8 % GS_D
    auto step = [&](auto s8_c, int b3, bool last) {
        constexpr int s8 = decltype(s8_c)::value, slot = s8 % GS_D;
        using SL = std::integral_constant<int, slot>;
        __builtin_amdgcn_sched_barrier(0);
        if (s8 < GS_D) {
            // (first block of a tile: the previous tile's epilogue put at least 8 stores -- they count too on gfx9 -- between
            // the rows and this block's DMAs; on the very first tile the prologue has already waited for everything)
            if (b3 == 0) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        pin_rows(SL{});
        f16x8 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[mi] = cvt_f16_rtn(ring[slot][mi][0], ring[slot][mi][1]);
        asm volatile("" :: "v"(af[0]), "v"(af[1]));      // (converted before the slot is reloaded)
        if (s8 == GS_NB - 3 && last && !(dbg & 64)) {    // this tile's inverse norms (uniform branch), retired by later waits
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const __attribute__((address_space(1))) float* p = invp + (size_t)row0 + mi * 16;
                gload16<0>(inv4[mi], (gptr)p);
            }
        }
        // rows of step + GS_D: of this block's tile, or (second half of a tile's last block) the first steps of the next tile
        if (s8 < GS_NB - GS_D) {
            load_rows(SL{}, rp0 + b3 * (GS_NB * 128), rp1 + b3 * (GS_NB * 128), std::integral_constant<int, (s8 + GS_D) * 128>{});
        } else {
            gptr q0 = last ? rn0 : rp0 + (b3 + 1) * (GS_NB * 128), q1 = last ? rn1 : rp1 + (b3 + 1) * (GS_NB * 128);
            load_rows(SL{}, q0, q1, std::integral_constant<int, (s8 + GS_D - GS_NB) * 128>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* bq = smem + pb * GS_BLK + s8 * GS_STEP + foffB;
#pragma unroll
        for (int n4 = 0; n4 < 2; ++n4) {
            f16x8 bh[4] = {};
            if (!(dbg & 16)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bh[i] = *(const f16x8*)(bq + (4 * n4 + i) * 1024);
            }
            if (dbg & 8) continue;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[mi][4 * n4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bh[i], acc[mi][4 * n4 + i], 0, 0, 0);
        }
    };
    for (int it = 0; it < n_mine; ++it) {
#pragma unroll 1
        for (int b3 = 0; b3 < GF_KT / GS_NB; ++b3) {
            const bool last = b3 == GF_KT / GS_NB - 1;
            stage_b(last ? 0 : b3 + 1, pb ^ 1);          // the next query block (the image repeats for every tile)
            step(std::integral_constant<int, 0>{}, b3, last);
            step(std::integral_constant<int, 1>{}, b3, last);
            step(std::integral_constant<int, 2>{}, b3, last);
            step(std::integral_constant<int, 3>{}, b3, last);
            step(std::integral_constant<int, 4>{}, b3, last);
            step(std::integral_constant<int, 5>{}, b3, last);
            step(std::integral_constant<int, 6>{}, b3, last);
            step(std::integral_constant<int, 7>{}, b3, last);
            // end of a query block: everyone is done with it, and the next has landed (its DMAs are older than row loads
            // this wave has already waited for)
            wg_barrier();
            pb ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); // the inverse norms (older than the last three steps' row loads)
        pin2(inv4[0], inv4[1]);
        // ---- epilogue: accumulator (mi, ni)[rr] = row 32 w + mi 16 + 4 lg + rr of the tile, query ni 16 + li16 ----
        const int n_valid = row_end - row0;
        int col_e = li16;
        asm volatile("" : "+v"(col_e));
        float cmax[8];
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) cmax[ni] = NEG_INF;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int blk = w * 32 + mi * 16;
            if (blk >= n_valid || (dbg & 32)) {         // wave-uniform: a fragment wholly behind the tile.  Its accumulators hold
#pragma unroll                                          // dot products of the NEXT tile's rows (or of the clamped last row):
                for (int ni = 0; ni < 8; ++ni)          // mask them, the emission loop below looks at both fragments
                    acc[mi][ni] = (f32x4){NEG_INF, NEG_INF, NEG_INF, NEG_INF};
                continue;
            }
            const int rb = blk + 4 * lg;
            const bool part = blk + 16 > n_valid;       // wave-uniform
            const f32x4 inv = inv4[mi];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                f32x4 v = acc[mi][ni] * inv;            // cosine = <e, q^> / ||e||
                if (part) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (rb + rr >= n_valid) v[rr] = NEG_INF;
                }
                acc[mi][ni] = v;                        // (kept for the emission pass below)
                cmax[ni] = max2_raw(cmax[ni], max3_raw(max2_raw(v[0], v[1]), v[2], v[3]));
            }
        }
        if (EMIT) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                if (__ballot(cmax[ni] >= thrv[ni]) == 0) continue;
                const int q = a.q_base + ni * 16 + col_e;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int rb = w * 32 + mi * 16 + 4 * lg;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float x = acc[mi][ni][rr];
                        const bool hit = x >= thrv[ni];                       // (rows behind the tile hold -inf: masked in the epilogue)
                        const unsigned long long hm = __ballot(hit);
                        if (hm != 0) {
                            const int pos = wave_cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                            if (hit && pos < a.wv_cap)
                                wvbuf[pos] = make_int4(row0 + rb + rr, q, __float_as_int(x), tile_of(jt));
                            wave_cnt += __popcll(hm);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            float m = cmax[ni];
            auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = max2_raw(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
            auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = max2_raw(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
            if (lg == 0) a.tmax_t[((size_t)jt * 8 + w) * 128 + ni * 16 + col_e] = m;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        jt = jn;
        row0 = row0n;
        row_end = a.tile_row[tile_of(jt) + 1];
        jn = jt + G < a.t_count ? jt + G : jt;
        row0n = a.tile_row[tile_of(jn)];
        rp0 = rn0; rp1 = rn1;
        rn0 = row_ptr(row0n, 0); rn1 = row_ptr(row0n, 1);
    }
    wait_vm0();                                          // (the prefetched block and rows of a tile that does not exist)
    if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = wave_cnt;
}

// ---- 256 queries per pass --------------------------------------------------------------------------------------------------
// The pass above moves 15.36 GB for 128 queries at ~0.9 of what a copy achieves: the bytes per query are the only lever left.
// This kernel serves 256 queries with the same bytes and the same shape -- 8 waves, wave w owns rows 32 w .. +32 of the tile
// and ALL queries, its rows never pass through LDS (ring of 4 K steps, 128 KB per CU in flight).  What changes: 128
// accumulator registers per wave instead of 64 (they take the accumulation half of the wave's 256 registers; the row
// ring, the fragments and the addresses the other half: thresholds and inverse row norms therefore live in LDS and rows are
// addressed scalar base + one 32-bit offset register); the f16 query image of a K step is 16 KB, so a block is 4 K steps
// (64 KB), double buffered, LDS-DMA one block ahead, ONE barrier per block (6 per tile).  The matrix cores see twice the
// work per byte (2 x 768 x 256 flop per row: ~1 PFLOP/s at the pass' byte rate, well under the f16 peak); the fragment
// reads are 128 KB of LDS per K step and CU (half of what the LDS delivers in the time HBM needs for the step's rows).
constexpr int G2_THREADS = 512;
constexpr int G2_NB = 4;                        // K steps per query block (= the depth of the row ring)
constexpr int G2_STEP = 16384;                  // bytes of one K step of the image: 256 queries x 64 B
constexpr int G2_BLK = G2_NB * G2_STEP;         // 64 KB
constexpr int G2_TMX = 2 * G2_BLK + 1024 + 8 * 256;   // offset of the tile maxima (below)
constexpr int G2_LDS = G2_TMX + 2 * 1024;             // + the 256 emission thresholds + 64 inverse row norms per wave + 2 x 256 tile maxima
// tile maxima are joined across the workgroup's waves in LDS as unsigned keys that order like the floats (atomic max on words)
__device__ __forceinline__ uint32_t ord_key(float x) {
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_val(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// a 16-byte global load in the scalar-base + 32-bit lane-offset form: one register per row pointer instead of two
template <int OFF, bool NT = false>
__device__ __forceinline__ void gload16s(f32x4& r, uint32_t voff, uint64_t sbase) {
    // NT: the non-temporal cache policy -- rows that are read once do not displace the query image (re-read by every
    // workgroup for every tile) from the L2
    if (NT) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// A copy the compiler cannot see through: the fragment of a bf16 row IS the loaded register, and a plain copy would let the
// register allocator rename the ring slots around the loop with moves of registers whose loads are still in flight (the
// loads are asynchronous behind its back).  With an opaque copy a slot is dead from here to its reload, like after the f32
// rows' conversion, and keeps its registers.
__device__ __forceinline__ f16x8 take16(const f32x4& r) {
    f32x4 o;
    asm volatile("v_mov_b32 %0, %1" : "=v"(o[0]) : "v"(r[0]));
    asm volatile("v_mov_b32 %0, %1" : "=v"(o[1]) : "v"(r[1]));
    asm volatile("v_mov_b32 %0, %1" : "=v"(o[2]) : "v"(r[2]));
    asm volatile("v_mov_b32 %0, %1" : "=v"(o[3]) : "v"(r[3]));
    return __builtin_bit_cast(f16x8, o);
}
// the lane id, computed where it is needed and opaque to the compiler (not hoisted, not kept in a register)
__device__ __forceinline__ uint32_t fresh_lane() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// a 16-byte LDS read the compiler does not wait for (the caller places s_waitcnt lgkmcnt and re-defines the register)
typedef __attribute__((address_space(3))) char lds_char;
template <int OFF>
__device__ __forceinline__ void lds_read16(f16x8& r, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t u) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// BF16 = true: the rows are a bf16 image of UNIT rows (1536 B each; msr_gemm.hip's candidate pass for 1024 queries per call):
// the fragments come out of the row ring as they are (no conversion, no inverse norms), a ring slot (128 bytes of a row) feeds
// two K steps, and a.nt = 4 query groups of 256 share the rows: the workgroups are dealt so that the nt
// workgroups with the same tile sequence have the same blockIdx % 8 (one XCD under round-robin dispatch: speed only) -- the
// rows come from HBM once per call and from that XCD's L2 for the other groups.
//
// TILED = true (f32 rows): the rows come from a copy of the matrix in FRAGMENT ORDER (tile_rows_kernel below): for every
// group of 16 rows and K step, the 2 KB the wave's two loads of that fragment and step take -- piece (lane, half) at
// half * 1024 + lane * 16 -- are contiguous, and every tile starts at a multiple of 16 rows of the copy.  A load instruction
// then reads 1 KB = 8 whole cache lines of its own; on the row-major matrix it reads 64 of the 128 bytes of 16 lines, the
// other halves follow in the next instruction and find their lines pending in the L1 (TCP_READ_TAGCONFLICT_STALL_CYCLES:
// 23 % of the L1's cycles).
// F16I = true (with BF16 = true, which stands for the data path of a row-major 16-bit image): the image holds the f32 rows
// AS THE f32 PASS CONVERTS THEM (f16, round to nearest, not normalised) -- the products, thresholds, margins and emitted entries
// are those of the f32 rows' pass bit for bit; only the conversion is not done again by every query group of a launch.  f16
// matrix instructions and the rows' inverse norms in the epilogue as on the f32 rows, the waves' tile maxima joined in LDS as
// on the bf16 image.
template <bool EMIT, bool BF16, bool TILED = false, bool NTL = false, bool F16I = false>
__global__ __launch_bounds__(G2_THREADS) void gemm_stream256_kernel(GemmF32Args a) {
    static_assert(!F16I || (BF16 && !TILED && !NTL), "the f16 image takes the 16-bit image's data path");
    // (A fragment-order copy of the bf16 image was measured too: 13.71 vs 13.97 ms per 1024 queries x 10 M rows, but 1.33 x
    // instead of 1.21 x the rows in HBM reads at 5 M rows and 7.9 GB more memory: not kept.  The addressing below stays general.)
    static_assert(!(BF16 && TILED), "the fragment-order copy exists for the f32 rows only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = BF16 ? MSR_DIM * 2 : GF_ROWB;             // bytes per row
    // A ring slot holds 128 bytes of each of the wave's rows = one full cache line per row and load pair (lane (li16, lg)
    // takes the 32 bytes at lg * 32 with two 16-byte loads): 32 floats = ONE K step of f32 rows, 64 bf16 = TWO K steps of bf16
    // rows (first the lanes' lower 16 bytes, then the upper: the query image orders K the same way, build_qimg2_kernel).
    // (bf16 rows loaded 64 bytes per K step -- half a line per row and instruction -- ran the pass at 15.0 ms per 1024
    // queries x 10 M rows instead of [see DESIGN.md]; with 64 rows x 128 queries per wave, i.e. half the LDS fragment reads
    // but every row loaded by two waves: 17.9 ms.  The row loads' path through the L1, not LDS, is what the bf16 pass feels.)
    constexpr int KPS = BF16 ? 2 : 1;                              // K steps per ring slot
    constexpr int RSLOTS = G2_NB / KPS;                            // ring slots = one query block's worth of rows
    constexpr int NMI = 2, NNI = 16;                               // the wave's fragments: 32 rows x 256 queries
    constexpr int WROWS = 16 * NMI;                                // rows per wave
    static_assert((G2_THREADS / 64) * WROWS == MSR_STREAM256_TILE_ROWS, "the fragment-order copy is padded for this many rows per tile visit");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w, wc = 0;                                      // row group, query half (none)
    const int li16 = lane & 15, lg = lane >> 4;
    // which tiles, which query group: a.nt groups of 256 queries share a tile sequence (nt == 1: every workgroup its own)
    const int nt = a.nt;
    int G = (int)gridDim.x, gid = (int)blockIdx.x, grp = 0;
    bool active = true;
    if (nt > 1) {
        const int per_x = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
        const int gpx = per_x / nt, mg = li / nt;
        grp = li - mg * nt;
        G = __builtin_amdgcn_readfirstlane(8 * gpx);
        gid = __builtin_amdgcn_readfirstlane(xcd * gpx + mg);
        grp = __builtin_amdgcn_readfirstlane(grp);
        active = mg < gpx;
    }
    const int nq_pad = nt * 256;
    int wave_cnt = EMIT && a.append ? __builtin_amdgcn_readfirstlane(a.wv_count[blockIdx.x * 8 + w]) : 0;
    int4* wvbuf = EMIT ? (int4*)a.wvbuf + ((size_t)blockIdx.x * 8 + w) * a.wv_cap : nullptr;
    if (!active || gid >= a.t_count) {                  // workgroup-uniform
        if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = wave_cnt;
        return;
    }
    // (the division runs on the vector unit; without the readfirstlane the tile counters and everything derived from them --
    // tile numbers, the tile table's loads -- stay in vector registers)
    const int n_mine = __builtin_amdgcn_readfirstlane((a.t_count - gid + G - 1) / G);
    const char* qimg = a.qimg + (size_t)grp * (GF_KT * G2_STEP);
    const int q_base = a.q_base + grp * 256;
    float* thr_lds = (float*)(smem + 2 * G2_BLK);
    if (EMIT && tid < 256) thr_lds[tid] = a.thr[grp * 256 + tid];
    // f32 rows: the lane's 16 emission thresholds (queries ni 16 + lane & 15) stay in 8 registers for the whole kernel, as f16
    // pairs rounded DOWN (a lower threshold only emits a few more entries; the final bound is applied in f32 later).  Read
    // from LDS they cost the epilogue one LDS round trip per 16 queries -- 16 per tile and wave, on its critical path.
    // (bf16 rows: no registers to spare, 250 of 256 in use; there the next threshold is requested one block of queries ahead.)
    constexpr bool THR_REGS = EMIT && !BF16;
    uint32_t thr_h[THR_REGS ? NNI / 2 : 1];
    if (THR_REGS) {
        auto down_h = [](float x) -> uint32_t {             // largest f16 <= x (inf stays inf; NaN stays NaN: never emits)
            _Float16 h = (_Float16)x;
            uint16_t b = __builtin_bit_cast(uint16_t, h);
            if ((float)h > x) b = (b & 0x8000u) ? (uint16_t)(b + 1) : (b == 0 ? (uint16_t)0x8001u : (uint16_t)(b - 1));
            return b;
        };
#pragma unroll
        for (int j = 0; j < NNI / 2; ++j) {
            const float lo = a.thr[grp * 256 + (2 * j) * 16 + li16], hi = a.thr[grp * 256 + (2 * j + 1) * 16 + li16];
            thr_h[j] = down_h(lo) | (down_h(hi) << 16);
        }
    }
    // Tile maxima, bf16 rows (1024 queries per pass: eight rows of maxima per tile would be 640 MB of writes per pass): the
    // eight waves' maxima of a (tile, query) are joined in LDS (atomic max on order-preserving keys; two sets, by tile
    // parity) and written as ONE row of 256 per tile -- after the first barrier of the NEXT tile, which every wave passes
    // only with its epilogue behind it.  f32 rows (160 MB per 256-query pass, 1 % of its traffic): every wave stores its own
    // row as before -- the join costs the pass 1.3 % and saves the transposing kernel the same.  The f16 image of the rows
    // (launches of several query groups) joins like the bf16 image: the maximum of the same values either way.
    constexpr bool JOIN = BF16;
    uint32_t* tmx = (uint32_t*)(smem + G2_TMX);
    if (JOIN) tmx[tid] = 0u;                            // (512 threads, 2 x 256 keys; 0 orders below every float)
    auto flush_tmax = [&](int tile_j, int par) {        // wave w: queries 32 w .. + 31 of the tile whose maxima sit in set par
        const int ln = (int)fresh_lane();               // (not the kernel's `lane`: nothing here is worth a register held across the kernel)
        if (ln < 32) {
            uint32_t* slot = tmx + par * 256 + w * 32 + ln;
            a.tmax_t[(size_t)tile_j * nq_pad + grp * 256 + w * 32 + ln] = ord_val(*slot);
            *slot = 0u;
        }
    };
    // the wave's 32 inverse row norms of a tile go through LDS: one DMA in the tile's last block (all 64 lanes take part: 64
    // floats, the upper half belongs to the next wave's rows and is not used; inv_pad is padded by 512 entries)
    float* inv_lds = (float*)(smem + 2 * G2_BLK + 1024) + w * 64;
    // query fragments: query 16 ni + li16, logical 16 B chunk lg of its 64 B, stored at physical chunk lg ^ (((q >> 3) & 1) << 1)
    // one query block: 64 KB, linear; wave w moves 8 KB of it
    auto stage_b = [&](int blk, int pb) {
        const char* base = uniform_ptr(qimg + (size_t)blk * G2_BLK + (size_t)w * 8192);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (uint32_t)(i * 1024 + lane * 16)),
                                             (lds_void*)(smem + pb * G2_BLK + w * 8192 + i * 1024), 16, 0, 0);
    };
    f32x4 acc[NMI][NNI];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_of = [&](int j) { return a.t_first + j * a.t_stride; };
    // A tile's rows are addressed as (scalar base of the tile's first row) + (32-bit byte offset of the lane's row): fragment
    // mi = row WROWS wr + 16 mi + li16 of the tile, at its 32-byte piece lg of a K step; a row past the end of the matrix
    // re-reads the last row (masked in the epilogue).  A tile spans < 1 MB, so the offset fits 32 bits with room to spare.
    // (TILED: row0 here is the tile's first row IN THE COPY, a multiple of 16: 16-row group g of the copy starts at g * 24 * 2 KB
    // = 16 g rows * 3072 B, the same product)
    auto tile_base = [&](int row0) { return uniform_u64((uint64_t)a.E + (uint64_t)row0 * ROWB); };
    // The bf16 image is padded with 512 zero rows (msr_enable_bf16): nothing to clamp, the lane's offset is the same for
    // every tile and the fragments differ by a constant that goes into the scalar base -- one offset register instead of eight.
    constexpr int NV = (BF16 || TILED) ? 1 : NMI;
    auto row_off = [&](int row0, int mi) -> uint32_t {
        if (TILED) return (uint32_t)(lane * 16);
        if (BF16) return (uint32_t)((WROWS * wr + li16) * ROWB + lg * 32);
        int64_t r = (int64_t)row0 + WROWS * wr + 16 * mi + li16;
        if (r > a.n_rows - 1) r = a.n_rows - 1;
        return (uint32_t)((r - row0) * ROWB + lg * 32);
    };
    int jt = gid;
    int row0 = a.tile_row[tile_of(jt)], row_end = a.tile_row[tile_of(jt) + 1];
    int jn = jt + G < a.t_count ? jt + G : jt;
    int row0n = a.tile_row[tile_of(jn)];
    uint64_t bp = tile_base(TILED ? a.tile_trow[tile_of(jt)] : row0), bn = tile_base(TILED ? a.tile_trow[tile_of(jn)] : row0n);
    uint32_t vp[NV], vn[NV];
#pragma unroll
    for (int mi = 0; mi < NV; ++mi) { vp[mi] = row_off(row0, mi); vn[mi] = row_off(row0n, mi); }

    // The row ring is driven by hand (see gemm_stream_kernel): inline-asm loads, explicit s_waitcnt before every use; vmcnt
    // retires in order.  Issue order per wave: [8 DMAs of the next block] then per step [wait, convert, 4 row loads,
    // compute].  f32 rows: the rows of step s of a block were loaded during step s of the block BEFORE (ring depth = block
    // length = 4): younger than them are the loads of three steps (12) and one block's 8 DMAs = 20, whatever s is.  bf16
    // rows: slot p (K steps 2p, 2p + 1) is reloaded in step 2p + 1 and waited for in step 2p of the next block: younger are
    // the other slot's 4 loads and 8 DMAs = 12.  (Epilogue stores and the inverse-norm DMA only add younger operations:
    // waiting for fewer than are in flight is always safe.)
    static_assert(NMI == 2, "the wait counts below assume 4 row loads per ring slot");
    f32x4 ring[RSLOTS][NMI][2];                          // [slot][fragment][16-byte half]
    auto load_rows = [&](auto slot_c, uint64_t sb, const uint32_t* v, auto off_c) {
        constexpr int slot = decltype(slot_c)::value, off = decltype(off_c)::value;
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            if (TILED) {                                 // group 2 wr + mi of the tile, ring slot `slot` of the block at sb
                const uint64_t b = sb + (uint64_t)((2 * wr + mi) * ((GF_KT / KPS) * 2048) + slot * 2048);
                gload16s<0, NTL>(ring[slot][mi][0], v[0], b);
                gload16s<1024, NTL>(ring[slot][mi][1], v[0], b);
            } else {
                const uint64_t b = BF16 ? sb + (uint64_t)(mi * 16 * ROWB) : sb;
                gload16s<off, NTL>(ring[slot][mi][0], v[BF16 ? 0 : mi], b);
                gload16s<off + 16, NTL>(ring[slot][mi][1], v[BF16 ? 0 : mi], b);
            }
        }
    };
    auto pin_rows = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        pin4(ring[slot][0][0], ring[slot][0][1], ring[slot][1][0], ring[slot][1][1]);
    };
    // (Tried: waves 4 .. 7 half a tile behind waves 0 .. 3 -- K blocks in the order 3 4 5 0 1 2 -- so that a wave's epilogue
    // runs beside its SIMD partner's MFMAs and loads instead of beside its epilogue: 2 % SLOWER on the f32 rows, 5 % on the
    // bf16 rows.  The eight waves share every query block and its barrier; a late wave holds the others up either way.)
    constexpr int NKB = GF_KT / G2_NB;                   // 6 K blocks per tile
    constexpr int BLKB = TILED ? RSLOTS * 2048 : RSLOTS * 128;    // address step of a K block (row-major: bytes of a row)
    // ---- prologue: query block 0, rows of K block 0 ----
    stage_b(0, 0);
#pragma unroll
    for (int s = 0; s < RSLOTS; ++s)
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) ring[s][mi][0] = ring[s][mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_rows(std::integral_constant<int, 0>{}, bp, vp, std::integral_constant<int, 0>{});
    load_rows(std::integral_constant<int, 1>{}, bp, vp, std::integral_constant<int, 128>{});
    if (!BF16) {
        load_rows(std::integral_constant<int, RSLOTS - 2>{}, bp, vp, std::integral_constant<int, 256>{});
        load_rows(std::integral_constant<int, RSLOTS - 1>{}, bp, vp, std::integral_constant<int, 384>{});
    }
    wait_vm0();
    pin_rows(std::integral_constant<int, 0>{}); pin_rows(std::integral_constant<int, 1>{});
    pin_rows(std::integral_constant<int, RSLOTS - 2>{}); pin_rows(std::integral_constant<int, RSLOTS - 1>{});
    wg_barrier();
    int pb = 0;                                          // LDS buffer of the current query block

    const float NEG_INF = -__builtin_inff();
    f16x8 bh[4];                                         // query fragments in flight (see step)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) bh[i][j] = (_Float16)0.f;
    // one K step; S4: position in the query block = ring slot (compile time)
    auto step = [&](auto s4_c, int kn, bool last) {     // kn: the K block after this one (same tile)
        constexpr int s4 = decltype(s4_c)::value;
        constexpr int slot = s4 / KPS, half = s4 % KPS;  // (bf16 rows: K step s4 = half `half` of slot s4 / 2)
        using SL = std::integral_constant<int, slot>;
        __builtin_amdgcn_sched_barrier(0);
#ifdef MSR_DIAG
        if (a.dbg & 32768) __builtin_amdgcn_s_setprio(3);   // timing experiment: the wave's memory section above its partner's MFMAs
#endif
        if (half == 0) {
            if (BF16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            pin_rows(SL{});
        }
        // f32 rows: converted out of the slot.  bf16 rows: the loaded registers ARE the fragments -- taken out with an opaque
        // copy (take16).  (Reading them in place and reloading the slot behind the step's MFMAs was tried: the register
        // allocator then spills ring slots.)
        f16x8 af[NMI];
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
            af[mi] = BF16 ? take16(ring[slot][mi][half]) : cvt_f16_rtn(ring[slot][mi][0], ring[slot][mi][1]);
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) asm volatile("" :: "v"(af[mi]));     // (taken out of the slot before it is reloaded)
        if ((!BF16 || F16I) && s4 == 1 && last) {        // this tile's inverse norms (uniform branch), retired by later waits
            const char* src = uniform_ptr((const char*)(a.inv_pad + (size_t)row0 + w * 32));
            __builtin_amdgcn_global_load_lds((glb_void*)(src + (uint32_t)(fresh_lane() * 4)), (lds_void*)inv_lds, 4, 0, 0);
        }
        // rows of the same step of the NEXT block: of this tile, or (a tile's last block) of the next tile's first block
        if (half == KPS - 1) {                           // (the slot's last K step: everything has been taken out of it)
#ifdef MSR_DIAG
            if (!(a.dbg & 262144))                       // timing experiment (wrong results): no row loads behind the prologue
#endif
            {
            if (last) load_rows(SL{}, bn, vn, std::integral_constant<int, slot * 128>{});
            else load_rows(SL{}, bp + (uint64_t)(kn * BLKB), vp, std::integral_constant<int, slot * 128>{});
            }
        }
#ifdef MSR_DIAG
        if (a.dbg & 32768) __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // (the lane's fragment offset is recomputed from a fresh lane id in every step -- 7 instructions beside 32 MFMAs --
        // so that it is not one more value held in a register across the whole kernel: the allocator has none to spare, and
        // what it spills it reloads behind an s_waitcnt vmcnt(0) in the middle of the row pipeline)
        const uint32_t ln = fresh_lane();
        const uint32_t fo = (ln & 15) * 64 + (((ln >> 4) ^ (((ln >> 3) & 1) << 1)) << 4);
        const char* bq = smem + pb * G2_BLK + s4 * G2_STEP + wc * (128 * 64) + fo;
        // The query fragments roll through four registers sets: fragment f is read into set f % 4 right behind the two
        // MFMAs of fragment f - 4, i.e. six MFMAs (~100 cycles) before its own -- the LDS latency hides behind this wave's
        // own MFMAs instead of counting on the SIMD's other wave.  The first four of a step are read by the step before
        // (same query block); a block's first step reads its own.
        // (bf16 rows only: the f32 rows' pass waits for HBM, not for LDS, and has no registers for fragments that stay
        // live across steps)
        if (!BF16) {
#ifdef MSR_DIAG
            if (a.dbg & 8) return;                       // timing experiments (wrong results): rows and query blocks move, nothing is multiplied
#endif
#pragma unroll
            for (int n4 = 0; n4 < NNI / 4; ++n4) {
                f16x8 b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) b4[i] = *(const f16x8*)(bq + (4 * n4 + i) * 1024);
#ifdef MSR_DIAG
                if (a.dbg & 16) {                        // timing experiments: fragment reads, no MFMA
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(b4[i]));
                    continue;
                }
#endif
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[mi][4 * n4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], b4[i], acc[mi][4 * n4 + i], 0, 0, 0);
            }
            return;
        }
        // The reads are inline asm with hand-placed s_waitcnt lgkmcnt, like the row ring (the compiler, left to itself, waits
        // for every read right behind it): LDS reads return in order, four are in flight at every wait but the last three of
        // a block.
        const uint32_t la = (uint32_t)(uintptr_t)(lds_char*)bq;
        if (s4 == 0) {
            lds_read16<0>(bh[0], la); lds_read16<1024>(bh[1], la); lds_read16<2048>(bh[2], la); lds_read16<3072>(bh[3], la);
        }
        auto frag = [&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            constexpr int ahead = (s4 + 1 < G2_NB) ? 3 : (NNI - 1 - f < 3 ? NNI - 1 - f : 3);   // reads behind fragment f's
            if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(bh[f & 3]));
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
                acc[mi][f] = F16I ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bh[f & 3], acc[mi][f], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, af[mi]), __builtin_bit_cast(bf16x8, bh[f & 3]), acc[mi][f], 0, 0, 0);
            if (f + 4 < NNI) lds_read16<(f + 4) * 1024>(bh[f & 3], la);
            else if (s4 + 1 < G2_NB) lds_read16<G2_STEP + (f + 4 - NNI) * 1024>(bh[f & 3], la);
        };
        frag(std::integral_constant<int, 0>{}); frag(std::integral_constant<int, 1>{});
        frag(std::integral_constant<int, 2>{}); frag(std::integral_constant<int, 3>{});
        frag(std::integral_constant<int, 4>{}); frag(std::integral_constant<int, 5>{});
        frag(std::integral_constant<int, 6>{}); frag(std::integral_constant<int, 7>{});
        frag(std::integral_constant<int, 8>{}); frag(std::integral_constant<int, 9>{});
        frag(std::integral_constant<int, 10>{}); frag(std::integral_constant<int, 11>{});
        frag(std::integral_constant<int, 12>{}); frag(std::integral_constant<int, 13>{});
        frag(std::integral_constant<int, 14>{}); frag(std::integral_constant<int, 15>{});
    };
    int jt_prev = jt;
    for (int it = 0; it < n_mine; ++it) {
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            const bool last = kb == NKB - 1;
            const int kn = last ? 0 : kb + 1;
#ifdef MSR_DIAG
            if (!(a.dbg & 524288))                       // timing experiment (wrong results): the query image is staged once
#endif
            stage_b(kn, pb ^ 1);                         // the next query block (the image repeats for every tile)
            step(std::integral_constant<int, 0>{}, kn, last);
            step(std::integral_constant<int, 1>{}, kn, last);
            step(std::integral_constant<int, 2>{}, kn, last);
            step(std::integral_constant<int, 3>{}, kn, last);
            // end of a query block: the next one has landed once at most the 16 (bf16 rows: 8) row loads issued behind its
            // DMAs are in flight; then everyone is done with this one
            if (BF16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#ifdef MSR_DIAG
            if (!(a.dbg & 131072))                       // timing experiment (wrong results): no rendezvous at the end of a query block
#endif
            wg_barrier();
            pb ^= 1;
            if (JOIN && kb == 0 && it > 0) flush_tmax(jt_prev, (it - 1) & 1);
        }
        // (the inverse norms were requested three steps ago and are older than the last 12 row loads; the lanes read what
        // their own wave's DMA wrote: no barrier.  Unit-row image: nothing to scale with)
        // (lane-derived values of the epilogue start from a fresh lane id: derived from the kernel's `lane` they are hoisted
        // out of the tile loop -- eight row numbers, an address -- and then spilled, to be reloaded one by one behind an
        // s_waitcnt vmcnt(0) at the start of every tile's epilogue)
        const int ln_e = (int)fresh_lane();
        const int lg_e = ln_e >> 4;
        f32x4 inv4[NMI];
        if (!BF16 || F16I) {
            // (16-bit image: the DMA is older than the 8 row loads of the block's steps 1 and 3, which the block's last wait left
            // in flight at most)
            if (F16I) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) inv4[mi] = *(const f32x4*)(inv_lds + mi * 16 + 4 * lg_e);
        }
        // ---- epilogue, one block of 16 queries at a time: accumulator (mi, ni)[rr] = row WROWS wr + mi 16 + 4 lg + rr of
        //      the tile, query 128 wc + ni 16 + li16 of the group ----
        const int n_valid = row_end - row0;
        const int col_e = wc * 128 + (ln_e & 15);
        float thr_nx = (EMIT && !THR_REGS) ? thr_lds[col_e] : 0.f;        // (bf16 rows: the threshold of the NEXT 16 queries is
        //                                                                    requested while this block's maxima are computed)
#ifdef MSR_DIAG
        if (a.dbg & 65536) {                             // timing experiment (wrong results): no epilogue at all
#pragma unroll
            for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else
#endif
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) {
            f32x4 v[NMI];
            float cmax = NEG_INF;
            const float thr_cur = thr_nx;
            if (EMIT && !THR_REGS && ni + 1 < NNI) thr_nx = thr_lds[(ni + 1) * 16 + col_e];
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) {
                const int blk = wr * WROWS + mi * 16;
                if (BF16 && !F16I) v[mi] = acc[mi][ni]; // unit rows: the cosine as it is
                else v[mi] = acc[mi][ni] * inv4[mi];    // cosine = <e, q^> / ||e||
                if (blk + 16 > n_valid) {               // (wave-uniform: only a tile's last, partly filled fragment -- or one wholly
#pragma unroll                                          // behind the tile -- pays for the per-row test)
                    for (int rr = 0; rr < 4; ++rr)      // rows behind the tile: products of the next tile's rows (or of the
                        if (blk + 4 * lg_e + rr >= n_valid) v[mi][rr] = NEG_INF;  // clamped last row) -- masked
                }
                cmax = max2_raw(cmax, max3_raw(max2_raw(v[mi][0], v[mi][1]), v[mi][2], v[mi][3]));
                acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (EMIT) {
                float thr;
                if (THR_REGS) thr = (float)__builtin_bit_cast(_Float16, (uint16_t)(ni & 1 ? thr_h[ni >> 1] >> 16 : thr_h[ni >> 1] & 0xffffu));
                else thr = thr_cur;
                if (__ballot(cmax >= thr) != 0) {
                    const int q = q_base + ni * 16 + col_e;
#pragma unroll
                    for (int mi = 0; mi < NMI; ++mi) {
                        const int rb = wr * WROWS + mi * 16 + 4 * lg_e;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const float x = v[mi][rr];
                            const bool hit = x >= thr;
                            const unsigned long long hm = __ballot(hit);
                            if (hm != 0) {
                                const int pos = wave_cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                                if (hit && pos < a.wv_cap)
                                    wvbuf[pos] = make_int4(row0 + rb + rr, q, __float_as_int(x), tile_of(jt));
                                wave_cnt += __popcll(hm);
                            }
                        }
                    }
                }
            }
            float m = cmax;
            auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = max2_raw(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
            auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = max2_raw(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
            if (JOIN) {
                if (lg_e == 0) atomicMax(tmx + (it & 1) * 256 + ni * 16 + col_e, ord_key(m));
            } else {
                if (lg_e == 0) a.tmax_t[((size_t)jt * 8 + w) * nq_pad + grp * 256 + ni * 16 + col_e] = m;
            }
        }
        jt_prev = jt;
        jt = jn;
        row0 = row0n;
        row_end = a.tile_row[tile_of(jt) + 1];
        jn = jt + G < a.t_count ? jt + G : jt;
        row0n = a.tile_row[tile_of(jn)];
        bp = bn; bn = tile_base(TILED ? a.tile_trow[tile_of(jn)] : row0n);
#pragma unroll
        for (int mi = 0; mi < NV; ++mi) { vp[mi] = vn[mi]; vn[mi] = row_off(row0n, mi); }
    }
    wait_vm0();                                          // (the prefetched block and rows of a tile that does not exist)
    if (JOIN) {
        wg_barrier();                                    // (every wave's last epilogue is behind it)
        flush_tmax(jt_prev, (n_mine - 1) & 1);
    }
    if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = wave_cnt;
}

// query image of the 256-query kernel: [kt 24][q 256][physical chunk c' 4] x 16 B, group blockIdx.y = queries 256 g .. + 255
// BF16: the K order of the 16-bit image's data path; TOBF16: bf16 elements (else f16)
template <bool BF16, bool TOBF16 = BF16>
__global__ __launch_bounds__(256) void build_qimg2_kernel(const float* __restrict__ qn, int nq, f16x8* __restrict__ qimg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= GF_KT * 256 * 4) return;
    const int g = blockIdx.y;
    const int cp = i & 3, ql = (i >> 2) & 255, kt = i >> 10;
    const int q = 256 * g + ql;
    const int c = cp ^ (((ql >> 3) & 1) << 1);
    f16x8 h;
    if (q < nq) {
        // f16 image: chunk c of K step kt = K elements 32 kt + 8 c ..; bf16 image: the kernel takes K steps 2 p and 2 p + 1
        // from the lower / upper 16 bytes of the 32 bytes a lane loads at lg * 32 of a row's 128-byte line
        const float* src = qn + (size_t)q * MSR_DIM + (BF16 ? 64 * (kt >> 1) + 16 * c + 8 * (kt & 1) : 32 * kt + 8 * c);
        if (TOBF16) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (__bf16)src[j];
            h = __builtin_bit_cast(f16x8, b);
        } else {
            h = cvt_f16_rtn(*(const f32x4*)src, *(const f32x4*)(src + 4));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)0.f;     // (all-zero bits are 0.0 in both formats)
    }
    qimg[(size_t)g * (GF_KT * 256 * 4) + i] = h;
}

// query image of the streaming kernel: [kt 24][q 128][physical chunk c' 4] x 16 B = f16 of dims 32 kt + 8 c .. + 8 of the
// normalised query q, c = c' ^ (((q >> 3) & 1) << 1); queries >= nq are zero
__global__ __launch_bounds__(256) void build_qimg1_kernel(const float* __restrict__ qn, int nq, f16x8* __restrict__ qimg) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // (kt, q, c') of group blockIdx.y (queries 128 g .. 128 g + 127)
    if (i >= GF_KT * 128 * 4) return;
    const int g = blockIdx.y;
    const int cp = i & 3, ql = (i >> 2) & 127, kt = i >> 9;
    const int q = 128 * g + ql;
    const int c = cp ^ (((ql >> 3) & 1) << 1);
    f16x8 h;
    if (q < nq) {
        const float* src = qn + (size_t)q * MSR_DIM + 32 * kt + 8 * c;
        h = cvt_f16_rtn(*(const f32x4*)src, *(const f32x4*)(src + 4));
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)0.f;
    }
    qimg[(size_t)g * (GF_KT * 128 * 4) + i] = h;
}

__global__ __launch_bounds__(256) void pad_inv_kernel(const float* __restrict__ inv, int64_t n, int64_t n_pad, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) out[i] = i < n ? inv[i] : 1.0f;
}

__global__ __launch_bounds__(256) void f16_row_error_kernel(const float* __restrict__ src, const float* __restrict__ inv_norm,
                                                             int64_t n_rows, uint32_t* __restrict__ err_max) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    float worst = 0.f;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const f32x4* p = (const f32x4*)(src + (size_t)r * MSR_DIM);
        float ss = 0.f;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const f32x4 x = p[lane + 64 * h];
            ss += f16_err2(x.x) + f16_err2(x.y) + f16_err2(x.z) + f16_err2(x.w);
        }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        worst = fmaxf(worst, sqrtf(ss) * inv_norm[r]);
    }
    if (lane == 0) atomicMax(err_max, __float_as_uint(worst));
}

// margin[q] = 2 eps_q + slack, eps_q = dE (1 + dq) + dq with dq = ||q^ - f16(q^)|| (q^: the normalised query); the slack
// covers the f32 accumulation of 768 products on both sides of the argument (msr_batch.hip)
__global__ __launch_bounds__(256) void f16_margin_kernel(const float* __restrict__ qn, int nq, const uint32_t* __restrict__ err_max,
                                                          float* __restrict__ margin) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    float ss = 0.f;
    for (int j = lane; j < MSR_DIM; j += 64) ss += f16_err2(qn[(size_t)q * MSR_DIM + j]);
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) {
        const float dq = sqrtf(ss) * 1.0001f, dE = __uint_as_float(*err_max) * 1.0001f;
        margin[q] = 2.0f * (dE * (1.0f + dq) + dq * 1.000001f) + 1.0e-4f;
    }
}

// One workgroup per query: entries (row, approximate score) -> the distinct documents they belong to = the candidates, each
// with the run of ITS emitted rows: the entries are sorted by (document, row) and the rows written back over pairs[q][.].x in
// that order; cand_first_len[slot] = first | len << 13 names the candidate's run.  msr_batch_rescore_rows computes the exact
// f32 cosines of exactly those rows.  That is enough: a row whose exact cosine is >= the exact k-th document score sigma has
// an approximate score >= sigma - eps >= thr2, so it was emitted -- for every document with max-cosine >= sigma the arg-max
// row (and every row tied with it) is among its emitted rows and the document's score comes out exact; a document below
// sigma gets a score <= its true one and stays below.  (Rescoring all rows of each candidate document, as rounds 2-4 did,
// read ~5 x the bytes.)  Overflow on the way: no candidates and gate[q / 64] |= 1 -- one
// gate word per slice of 64 queries, the unit of the caller's gated sweeps, so a query with a huge tie group sends its own
// slice back to the sweeps, not the whole call (an overflowing wave buffer concerns every query: all slices).
static_assert(GF_PAIR_CAP <= (1 << 13), "cand_first_len packs first and len into 13 bits each");
__global__ __launch_bounds__(1024) void gemm_f32_cand_kernel(int2* __restrict__ pairs, int32_t* __restrict__ pair_n,
                                                              const int32_t* __restrict__ chunk_doc, int64_t n_rows,
                                                              const int32_t* __restrict__ wv_count, int n_waves, int wv_cap,
                                                              const int32_t* __restrict__ flag, int32_t* __restrict__ cand_doc,
                                                              int32_t* __restrict__ cand_first_len,
                                                              int32_t* __restrict__ cand_n, int32_t* __restrict__ gate) {
    __shared__ uint64_t key[GF_PAIR_CAP];               // (document + 1) << 32 | row; 0 = pad / an entry that names no row
    __shared__ int s_over, s_n;
    const int q = blockIdx.x, t = threadIdx.x;
    const int raw = pair_n[q];
    int over = raw > GF_PAIR_CAP || flag[q];
    for (int i = t; i < n_waves; i += 1024) over |= wv_count[i] > wv_cap;
    if (t == 0) { s_n = 0; s_over = 0; }
    __syncthreads();
    if (over) s_over = 1;
    __syncthreads();
    if (s_over) {
        if (t == 0) { cand_n[q] = 0; pair_n[q] = 0; atomicOr(gate + (q >> 6), 1); }   // the gate of the query's 64-query slice
        return;
    }
    int P = 64;
    while (P < raw) P <<= 1;
    for (int i = t; i < P; i += 1024) {
        uint64_t kk = 0u;                               // (pad key)
        if (i < raw) {
            const int64_t row = pairs[(size_t)q * GF_PAIR_CAP + i].x;
            if (row >= 0 && row < n_rows)               // (a row index is never trusted as an address)
                kk = ((uint64_t)((uint32_t)chunk_doc[row] + 1u) << 32) | (uint32_t)row;
        }
        key[i] = kk;
    }
    __syncthreads();
    for (int kk = 2; kk <= P; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int idx = t; idx < (P >> 1); idx += 1024) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int p = i | j;
                const bool desc = (i & kk) == 0;
                const uint64_t a = key[i], b = key[p];
                if (desc ? a < b : a > b) { key[i] = b; key[p] = a; }
            }
            __syncthreads();
        }
    // (descending: the pads end up behind the entries; a document's rows are adjacent, highest row first)
    for (int i = t; i < P; i += 1024) {
        const uint64_t kk = key[i];
        if (kk == 0) continue;
        pairs[(size_t)q * GF_PAIR_CAP + i].x = (int32_t)(uint32_t)kk;
        const uint32_t d = (uint32_t)(kk >> 32);
        if (i == 0 || (uint32_t)(key[i - 1] >> 32) != d) {
            int len = 1;
            while (i + len < P && (uint32_t)(key[i + len] >> 32) == d) ++len;
            const int slot = atomicAdd(&s_n, 1);
            cand_doc[(size_t)q * MSR_SEL_CAP + slot] = (int32_t)(d - 1u);
            cand_first_len[(size_t)q * MSR_SEL_CAP + slot] = i | (len << 13);
        }
    }
    __syncthreads();
    if (t == 0) { cand_n[q] = s_n; pair_n[q] = 0; }
}

int g_f32_dbg = 0;

template <bool EMIT>
hipError_t launch_stream_t(const GemmF32Args& a, int grid, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t err = hipFuncSetAttribute((const void*)gemm_stream_kernel<EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS);
        if (err != hipSuccess) return err;
        attr_done = true;
    }
    gemm_stream_kernel<EMIT><<<grid, GF_THREADS, GS_LDS, stream>>>(a);
    return hipGetLastError();
}
template <bool EMIT, bool BF16, bool TILED = false, bool NTL = false, bool F16I = false>
hipError_t launch_stream256_t(const GemmF32Args& a, int grid, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t err = hipFuncSetAttribute((const void*)gemm_stream256_kernel<EMIT, BF16, TILED, NTL, F16I>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
        if (err != hipSuccess) return err;
        attr_done = true;
    }
    gemm_stream256_kernel<EMIT, BF16, TILED, NTL, F16I><<<grid, G2_THREADS, G2_LDS, stream>>>(a);
    return hipGetLastError();
}

// dst[r] = f16(src[r]) (round to nearest: what the f32 rows' pass does in registers) for r < n_rows, zero rows up to n_pad
__global__ __launch_bounds__(256) void f16_rows_kernel(const float* __restrict__ src, int64_t n_rows, int64_t n_pad,
                                                        f16x8* __restrict__ dst) {
    const int64_t n8 = n_pad * (MSR_DIM / 8), stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const int64_t r = i / (MSR_DIM / 8);
        f16x8 h;
        if (r < n_rows) {
            const float* p = src + i * 8;
            h = cvt_f16_rtn(*(const f32x4*)p, *(const f32x4*)(p + 4));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)0.f;
        }
        dst[i] = h;
    }
}

// The fragment-order copy of the f32 rows (see gemm_stream256_kernel, TILED): one workgroup per tile; piece (lane, half) of
// (16-row group g, K step t) of the copy = floats 32 t + 8 (lane >> 4) + 4 half .. + 3 of row 16 g + (lane & 15) of the tile.
// Rows of a tile's last group behind the tile's end are zero.
// ROWB: bytes of a row (3072: f32 rows; 1536: the bf16 image); a row has ROWB / 128 slots of 128 bytes.
template <int ROWB>
__global__ __launch_bounds__(256) void tile_rows_kernel(const char* __restrict__ E, const int32_t* __restrict__ tile_row,
                                                         const int32_t* __restrict__ tile_trow, f32x4* __restrict__ Et) {
    constexpr int SLOTS = ROWB / 128;
    const int t_ = blockIdx.x;
    const int row0 = tile_row[t_], n = tile_row[t_ + 1] - row0;
    const int groups = (n + 15) >> 4;
    f32x4* dst = Et + (size_t)tile_trow[t_] * (ROWB / 16);
    const int pieces = groups * SLOTS * 128;             // 16-byte pieces of the tile's copy
    for (int i = threadIdx.x; i < pieces; i += 256) {
        const int lane = i & 63, half = (i >> 6) & 1, sl = (i >> 7) % SLOTS, g = i / (128 * SLOTS);
        const int lr = 16 * g + (lane & 15);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (lr < n) v = *(const f32x4*)(E + (size_t)(row0 + lr) * ROWB + 128 * sl + 32 * (lane >> 4) + 16 * half);
        dst[i] = v;
    }
}
// width: queries per pass (128 or 256; both kernels: 8 waves x 32 rows)
hipError_t launch_f32(int width, bool emit, const GemmF32Args& a, int grid, hipStream_t stream, bool tiled = false,
                      bool image16 = false) {
    if (image16)
        return emit ? launch_stream256_t<true, true, false, false, true>(a, grid, stream)
                    : launch_stream256_t<false, true, false, false, true>(a, grid, stream);
    // One query group per launch: every row is read exactly once, by one wave -- loaded with the non-temporal policy, so that the
    // 15 GB stream does not displace the 384 KB query image (re-read by every workgroup for every tile) from the L2s: 3.07 ->
    // 2.94 ms per pass on one box (profiles/r04_stream256_experiments.md).  Several groups per launch share the rows through the
    // L2 and keep the default policy.
    bool ntl = width == 256 && tiled && a.nt == 1;
#ifdef MSR_DIAG
    if (g_f32_dbg & 16384) ntl = false;                     // timing experiment: default cache policy for the rows
#endif
    if (ntl)
        return emit ? launch_stream256_t<true, false, true, true>(a, grid, stream) : launch_stream256_t<false, false, true, true>(a, grid, stream);
    if (width == 256 && tiled)
        return emit ? launch_stream256_t<true, false, true>(a, grid, stream) : launch_stream256_t<false, false, true>(a, grid, stream);
    if (width == 256) return emit ? launch_stream256_t<true, false>(a, grid, stream) : launch_stream256_t<false, false>(a, grid, stream);
    return emit ? launch_stream_t<true>(a, grid, stream) : launch_stream_t<false>(a, grid, stream);
}

}  // namespace

void msr_gemm_f32_set_dbg(int v) { g_f32_dbg = v; }

hipError_t msr_stream256_bf16_launch(bool emit, const StreamArgs& a, int grid, hipStream_t stream) {
    if (a.nt < 1 || (grid & 7) || (grid >> 3) < a.nt) return hipErrorInvalidValue;
#ifdef MSR_DIAG
    if (g_f32_dbg & (32768 | 65536 | 131072 | 262144 | 524288)) { StreamArgs b = a; b.dbg |= g_f32_dbg & (32768 | 65536 | 131072 | 262144 | 524288); return emit ? launch_stream256_t<true, true>(b, grid, stream) : launch_stream256_t<false, true>(b, grid, stream); }
#endif
    return emit ? launch_stream256_t<true, true>(a, grid, stream) : launch_stream256_t<false, true>(a, grid, stream);
}
hipError_t msr_stream256_bf16_qimage(const float* qn, int nq, int n_groups, void* qimg, hipStream_t stream) {
    build_qimg2_kernel<true><<<dim3((GF_KT * 256 * 4 + 255) / 256, n_groups), 256, 0, stream>>>(qn, nq, (f16x8*)qimg);
    return hipGetLastError();
}

hipError_t msr_tile_rows(const float* emb, const int32_t* tile_row, const int32_t* tile_trow, int n_tiles, void* emb_tiled,
                         hipStream_t stream) {
    if (n_tiles <= 0) return hipSuccess;
    tile_rows_kernel<MSR_DIM * 4><<<n_tiles, 256, 0, stream>>>((const char*)emb, tile_row, tile_trow, (f32x4*)emb_tiled);
    return hipGetLastError();
}


hipError_t msr_f16_rows(const float* emb, int64_t n_rows, int64_t n_pad, void* out, hipStream_t stream) {
    if (n_pad <= 0) return hipSuccess;
    f16_rows_kernel<<<8192, 256, 0, stream>>>(emb, n_rows, n_pad, (f16x8*)out);
    return hipGetLastError();
}

hipError_t msr_pad_inv_norm(const float* inv, int64_t n, int64_t n_pad, float* out, hipStream_t stream) {
    if (n_pad <= 0) return hipSuccess;
    pad_inv_kernel<<<(unsigned)((n_pad + 255) / 256), 256, 0, stream>>>(inv, n, n_pad, out);
    return hipGetLastError();
}

hipError_t msr_f16_row_error(const float* emb, const float* inv_norm, int64_t n_rows, uint32_t* err_max, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(err_max, 0, 4, stream);
    if (err != hipSuccess) return err;
    f16_row_error_kernel<<<8192, 256, 0, stream>>>(emb, inv_norm, n_rows, err_max);
    return hipGetLastError();
}

// Exact f32 top-k of up to 128 x g.max_groups queries; see the header of this file.  nq <= 128: one pass of the 128-query
// kernel; more: groups of 256 queries, one pass of the 256-query kernel each (a last group that is not full is padded with
// zero queries: the bytes are the same).  The passes of all groups are queued back to back; everything between and after
// them (the two selects over tile maxima, bucketing, candidate lists, rescoring, final sort) runs ONCE for all queries of
// the call.
// qn: [nq][768] normalised queries.  gate[s] != 0 (one word per 64 queries, zero on entry) when a query of slice s overflowed
// (the caller falls back for that slice).  ev (nullable): events around the sample pass (0, 1) and the emit pass (2, 3) of
// the first group.  *width_out (nullable): queries per pass of the kernel that ran.
hipError_t msr_gemm_f32_pass(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k, int k_part,
                             float* out_part, hipEvent_t* ev, int* width_out, hipStream_t stream) {
    const int W = nq > 128 ? 256 : 128;                 // queries per pass
    const int waves = 8;                                // waves per workgroup = emission buffers per workgroup (both kernels)
    const int G = (nq + W - 1) / W;
    if (nq <= 0 || G * W > 128 * g.max_groups || k < 1 || g.n_tiles < 2 * k) return hipErrorInvalidValue;
    if (width_out) *width_out = W;
    hipError_t err;
    // Launches of several query groups are bound by the matrix pipes and the vector issue beside them, not by HBM: they read
    // the f16 image of the rows when the engine holds one (the values the pass converts in registers otherwise -- same
    // products, same results; the conversion is not repeated by every group).  One group per launch: the f32 rows.
    int NT0 = 1;
    if (W == 256 && (g.n_cus & 7) == 0) NT0 = std::min(std::min(G, g.max_nt), g.n_cus >> 3);
    bool image16 = W == 256 && NT0 > 1 && g.emb_f16 != nullptr;
#ifdef MSR_DIAG
    if (g_f32_dbg & (2048 | 1048576)) image16 = false;      // timing experiments: one group per launch / the f32 rows
#endif
    if (image16) build_qimg2_kernel<true, false><<<dim3((GF_KT * 256 * 4 + 255) / 256, G), 256, 0, stream>>>(qn, nq, (f16x8*)g.qimg);
    else if (W == 256) build_qimg2_kernel<false><<<dim3((GF_KT * 256 * 4 + 255) / 256, G), 256, 0, stream>>>(qn, nq, (f16x8*)g.qimg);
    else build_qimg1_kernel<<<dim3((GF_KT * 128 * 4 + 255) / 256, G), 256, 0, stream>>>(qn, nq, (f16x8*)g.qimg);
    f16_margin_kernel<<<(nq + 3) / 4, 256, 0, stream>>>(qn, nq, g.err_max, g.margin);
    const float* margin = g.margin;
    int ss = g.n_tiles / (3 * k);                       // every ss-th tile bounds the k-th score from below: >= 3 k sampled tiles
    ss = ss < 1 ? 1 : (ss > 64 ? 64 : ss);          // (the weaker the bound, the more entries pass 2 emits: ~150 ss per query)
    const int n_s = (g.n_tiles - ss / 2 + ss - 1) / ss;
    const int grid = g.n_cus;
    const size_t qimg_bytes = (size_t)GF_KT * W * 64;   // one group's image
    // 256-query kernel: up to 4 groups per launch walk the same tile sequence on CUs of one XCD -- the rows come from HBM
    // once per launch and from that XCD's L2 for the other groups (the bytes per query are what the pass costs)
    int NT = 1;
    if (W == 256 && (grid & 7) == 0) NT = std::min(std::min(G, g.max_nt), grid >> 3);
#ifdef MSR_DIAG
    if (g_f32_dbg & 2048) NT = 1;                       // timing experiments: one group per launch
    if (g_f32_dbg & 4096) NT = std::min(NT, 2);
#endif
    GemmF32Args a{};
    a.dbg = g_f32_dbg; a.nt = 1;
    a.E = (const char*)ix.emb; a.inv_pad = g.inv_pad; a.tile_row = g.tile_row;
    bool tiled = W == 256 && g.emb_tiled != nullptr && !image16;
#ifdef MSR_DIAG
    if (g_f32_dbg & 8192) tiled = false;               // timing experiments: the row-major matrix
#endif
    if (tiled) { a.E = (const char*)g.emb_tiled; a.tile_trow = g.tile_trow; }
    if (image16) a.E = (const char*)g.emb_f16;
    const int tmax_parts = image16 ? 1 : waves;         // (the 16-bit image's data path joins the waves' tile maxima in LDS: one row per tile)
    a.n_rows = ix.n_chunks; a.tmax_t = g.tmax_t;
    // ---- pass 1 of every group: maxima of every ss-th tile -> emission thresholds ----
    a.t_first = ss / 2; a.t_stride = ss; a.t_count = n_s;
    for (int gi = 0; gi < G; gi += NT) {
        a.nt = std::min(NT, G - gi);
        a.qimg = (const char*)g.qimg + gi * qimg_bytes;
        if (ev && gi == 0 && (err = hipEventRecord(ev[0], stream)) != hipSuccess) return err;
        if ((err = launch_f32(W, false, a, grid, stream, tiled, image16)) != hipSuccess) return err;
        if (ev && gi == 0 && (err = hipEventRecord(ev[1], stream)) != hipSuccess) return err;
        if ((err = msr_gemm_tmax(g.tmax_t, n_s, tmax_parts, W * a.nt, g.tmax + (size_t)gi * W * g.tmax_stride, g.tmax_stride, stream)) != hipSuccess) return err;
    }
    if ((err = msr_gemm_kth(g.tmax, n_s, g.tmax_stride, nq, W * G, k, margin, g.thr, g.flag, stream)) != hipSuccess) return err;
    // ---- pass 2 of every group: maxima of all tiles + the entries at or above the threshold (one set of wave buffers) ----
    a.t_first = 0; a.t_stride = 1; a.t_count = g.n_tiles;
    a.wvbuf = g.wvbuf; a.wv_cap = g.wv_cap * (8 / waves); a.wv_count = g.wv_count;
    for (int gi = 0; gi < G; gi += NT) {
        a.nt = std::min(NT, G - gi);
        a.qimg = (const char*)g.qimg + gi * qimg_bytes;
        a.thr = g.thr + gi * W; a.q_base = gi * W; a.append = gi > 0;
        if (ev && gi == 0 && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
        if ((err = launch_f32(W, true, a, grid, stream, tiled, image16)) != hipSuccess) return err;
        if (ev && gi == 0 && (err = hipEventRecord(ev[3], stream)) != hipSuccess) return err;
        if ((err = msr_gemm_tmax(g.tmax_t, g.n_tiles, tmax_parts, W * a.nt, g.tmax + (size_t)gi * W * g.tmax_stride, g.tmax_stride, stream)) != hipSuccess) return err;
    }
    if ((err = msr_gemm_kth(g.tmax, g.n_tiles, g.tmax_stride, nq, W * G, k, margin, g.thr2, nullptr, stream)) != hipSuccess) return err;
    // (sharded callers: what this shard can vouch for towards the k-th score over all shards, see msr_internal.h)
    if (out_part && (err = msr_gemm_kth(g.tmax, g.n_tiles, g.tmax_stride, nq, nq, k_part, margin, out_part, nullptr, stream,
                                        -__builtin_inff(), 0.5f)) != hipSuccess) return err;
    return hipSuccess;
}

namespace {
// thr2[q] = max(thr2[q], bound[q] - margin[q] / 2): the threshold of this shard's own tile maxima, raised to what all shards
// together guarantee for the k-th score (bound is in exact-cosine space; a filter score is within margin / 2 of the exact one)
__global__ __launch_bounds__(256) void raise_thr_kernel(float* __restrict__ thr2, const float* __restrict__ bound,
                                                         const float* __restrict__ margin, int nq) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float b = bound[q] - 0.5f * margin[q];
    if (b > thr2[q]) thr2[q] = b;                           // (NaN / -inf bounds leave the local threshold alone)
}
}  // namespace

hipError_t msr_gemm_f32_finish(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k, const float* bound,
                               int32_t* out_doc, float* out_score, int32_t* out_chunk, int32_t* out_n, int32_t* gate,
                               hipStream_t stream) {
    const int waves = 8, grid = g.n_cus;
    const int wv_cap = g.wv_cap * (8 / waves);
    if (nq <= 0 || k < 1) return hipErrorInvalidValue;
    hipError_t err;
    if (bound) {
        raise_thr_kernel<<<(nq + 255) / 256, 256, 0, stream>>>(g.thr2, bound, g.margin, nq);
        if ((err = hipGetLastError()) != hipSuccess) return err;
    }
    if ((err = msr_gemm_bucket(g.wvbuf, wv_cap, g.wv_count, grid * waves, g.thr2, g.pairs, g.pair_n, stream)) != hipSuccess) return err;
    // candidates with the runs of their emitted rows (cand_chunk carries first | len << 13 in, the arg-max row out)
    gemm_f32_cand_kernel<<<nq, 1024, 0, stream>>>((int2*)g.pairs, g.pair_n, ix.chunk_doc, ix.n_chunks, g.wv_count, grid * waves,
                                                  wv_cap, g.flag, g.cand_doc, g.cand_chunk, g.cand_n, gate);
    if ((err = hipGetLastError()) != hipSuccess) return err;
    return msr_batch_rescore_rows(ix, qn, nq, k, (const int32_t*)g.pairs, 2, GF_PAIR_CAP, g.cand_doc, g.cand_score, g.cand_chunk,
                                  g.cand_n, out_doc, out_score, out_chunk, out_n, stream);
}

hipError_t msr_gemm_f32_topk(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k,
                             int32_t* out_doc, float* out_score, int32_t* out_chunk,
                             int32_t* out_n, int32_t* gate, hipEvent_t* ev, int* width_out, hipStream_t stream) {
    hipError_t err = msr_gemm_f32_pass(g, ix, qn, nq, k, k, nullptr, ev, width_out, stream);
    if (err != hipSuccess) return err;
    return msr_gemm_f32_finish(g, ix, qn, nq, k, nullptr, out_doc, out_score, out_chunk, out_n, gate, stream);
}
