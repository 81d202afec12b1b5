// K2 + K3 + K4 for batches of 65 .. 128 queries: the dense scan as a tiled GEMM over the ORIGINAL f32 rows (gfx950).
//
// Same cosine and per-document max as the sweeps (reference: reranker/reranker_api.py:285, :370; Retriever.quick_search,
// search_api.py:60,87), same arithmetic as the default sweep -- every f32 row value is split on the fly into two f16 pieces
// (x = hi + lo) and three v_mfma_f32_16x16x32_f16 (lo*hi + hi*lo + hi*hi) replace the f32 products, f32 accumulation;
// error bound 8e-6 on the cosine for row norms in [0.5, 2], DESIGN.md section 3 -- but organised like the bf16 candidate
// GEMM of msr_gemm.hip instead of the K-split sweep:
//   * one pass over E serves 128 queries (the sweep: 64), so a 128-query step reads the 15.36 GB once instead of twice;
//     3 x 2 x 768 flop per (row, query) on the f16 matrix pipe keep it about level with the HBM time of that one pass;
//   * no cross-wave reduction, no LDS ring of per-document maxima, no score rows: the epilogue only (a) stores the
//     tile maxima and (b) appends every (query, row, score) above a threshold to a per-wave buffer.  The threshold of a
//     query is a LOWER BOUND of its k-th largest per-document score: tiles are cut at document boundaries, so the k-th
//     largest tile maximum is attained by k different documents.  Pass 1 computes the maxima of every 16th tile (the bound
//     used for emission), pass 2 all of them (a much tighter bound used to thin the emitted entries afterwards).
//     The emitted scores ARE the final scores (no rescoring: they carry the default path's own arithmetic), so the bound
//     needs no margin: a document of the top-k has score >= t >= bound and is emitted.
//   * finish (gemm_f32_final_kernel, one workgroup per query): per-document maximum with its first arg-max row, exact sort by
//     (score desc, document asc), top-k.  A query whose entries do not fit (huge tie groups) is flagged; the caller then
//     runs the sweeps for the batch (msr_engine.hip gates them on that flag on the device, no host round trip).
//
// Kernel shape: persistent workgroup per CU, tile 256 rows x 128 queries, K step 32; wave w owns rows 32 w .. +32 and ALL
// 128 queries, so every f32 row fragment is split into its f16 pieces exactly once:
//   LDS  4 x 32 KB of rows (256 x 128 B, LDS-DMA) + 2 x 16 KB of query pieces (hi | lo) = all 160 KB; 16 B chunks of a row
//        XOR-swizzled so the 32 B fragment reads (two ds_read_b128) are conflict-free;
//   DMA  waves 0-3 issue the rows (of the step after next-but-one: THREE steps of flight, ~80 KB per CU in flight -- the
//        rows come from HBM and are read once), waves 4-7 the query pieces of the next step (L2-resident).  vmcnt is an
//        in-order counter per wave: with one kind of DMA per wave, "all but the 16 youngest" retires exactly the rows of the
//        next step, and the query waves simply wait for everything;
//   per step and wave: 4 + 16 fragment reads, 2 splits (VALU), 48 MFMAs, one wait, ONE barrier;
//   the inverse norms of a tile's rows (16 per lane) are fetched with plain loads two steps before the tile's last barrier
//   (inline asm: the compiler must not see an ordinary load next to in-flight DMAs, it would drain them).
#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"
#include "msr_frag.h"
#include "msr_gemm_dev.h"

namespace {

constexpr int GF_THREADS = 512;
constexpr int GF_KT = MSR_DIM / 32;             // 24 K steps per tile
constexpr int GF_ROWB = MSR_DIM * 4;            // bytes per f32 row
constexpr int GF_A = 32768, GF_B = 16384;       // bytes of one row buffer / one query buffer
constexpr int GF_NA = 4;                        // row buffers
constexpr int GF_LDS = GF_NA * GF_A + 2 * GF_B; // 160 KB
constexpr int GF_PAIR_CAP = 4096;

struct GemmF32Args {
    const char* E;             // f32 [n_rows][768] (caller's matrix: NOT padded, the last tile clamps its row index)
    const float* inv_pad;      // [n_rows + 512] inverse norms (engine-owned padded copy)
    const char* qimg;          // [24 K steps][hi | lo][128 queries][64 B] f16 pieces, chunk-swizzled (build_qimg_kernel)
    const int32_t* tile_row;   // [n_tiles + 1]
    int64_t n_rows;
    int t_first, t_stride, t_count;
    float* tmax_t;             // [t_count][8 waves][128]
    const float* thr;          // [128] emit threshold (+inf: never)                                   -- emit pass only
    int4* wvbuf; int wv_cap; int32_t* wv_count;   // per-wave emission buffers {row, query, score bits, tile}   -- emit pass only
    int dbg;                   // -DMSR_DIAG builds only (timing experiments, wrong results): bit 2 no DMA after the prologue,
                               // bit 3 no MFMA, bit 4 no fragment reads / splits
};

// a 64-bit value the compiler can prove wave-uniform (scalar registers): lets the DMA use the saddr + 32-bit voffset form
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

template <bool EMIT>
__global__ __launch_bounds__(GF_THREADS) void gemm_f32_kernel(GemmF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave w owns rows 32 w .. 32 w + 32 of the tile, ALL 128 queries
    const int li16 = lane & 15, lg = lane >> 4;
    const int G = (int)gridDim.x, gid = (int)blockIdx.x;
    int wave_cnt = 0;
    int4* wvbuf = EMIT ? a.wvbuf + ((size_t)blockIdx.x * 8 + w) * a.wv_cap : nullptr;
    if (gid >= a.t_count) {                             // workgroup-uniform
        if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = 0;
        return;
    }
    const int n_mine = (a.t_count - gid + G - 1) / G;

    // ---- per-lane constants ----
    // DMA of a row buffer (waves 0..3): instruction t (0..7) of wave w fills rows 8 (8 w + t) .. +8: lane -> (row R, physical
    // chunk lane & 7), which holds logical chunk c = (lane & 7) ^ f(R & 15), f(r) = ((r >> 1) & 3) << 1 | (r >> 3).
    // The per-lane part of the source address is the same for every tile: (R, chunk) -> a 32-bit offset from the tile's
    // first row; the tile / K step part is wave-uniform (scalar registers), so a DMA costs no address arithmetic.
    const int rowA0 = 64 * (w & 3) + (lane >> 3);      // + 8 t
    uint32_t offA[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
        offA[t] = (uint32_t)((rowA0 + 8 * t) * GF_ROWB + (((lane & 7) ^ ((((lane >> 4) & 3) << 1) | (t & 1))) * 16));
    // fragment reads: row li16 of a 16-row block, logical chunks 2 lg and 2 lg + 1 (32 B = 8 floats)
    const int fr = (((li16 >> 1) & 3) << 1) | ((li16 >> 3) & 1);
    uint32_t foffA[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) foffA[h] = (uint32_t)(li16 * 128 + (((2 * lg + h) ^ fr) << 4));
    const uint32_t foffB = (uint32_t)(li16 * 64 + ((lg ^ (((li16 >> 3) & 1) << 1)) << 4));
    const uint32_t a_base = (uint32_t)(w * 32 * 128);

    auto a_slot = [](int j) { return j * GF_A; };
    auto b_slot = [](int d) { return GF_NA * GF_A + d * GF_B; };
    // rows of the tile that starts at `row0`, K step kt -> row buffer (waves 0..3, 8 x 1 KiB each); `clamp`: the tile
    // sticks out of the matrix (only the last tile can): rows past the end re-read the last row (they are masked later)
    auto stage_a = [&](int row0, int kt, int slot, bool clamp) {
        const char* base = uniform_ptr(a.E + (size_t)row0 * GF_ROWB + (size_t)kt * 128);
#ifdef MSR_DIAG
        if (a.dbg & 128) {                              // timing experiment: the 32 KB of a K step as ONE contiguous block
            const char* lin = uniform_ptr(a.E + ((size_t)row0 * GF_ROWB / 32768 * 32768) + (size_t)kt * 32768 + (size_t)(8 * (w & 3)) * 1024);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                __builtin_amdgcn_global_load_lds((glb_void*)(lin + (uint32_t)(t * 1024 + lane * 16)),
                                                 (lds_void*)(smem + slot + (8 * (w & 3) + t) * 1024), 16, 0, 0);
            return;
        }
#endif
        if (!clamp) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                __builtin_amdgcn_global_load_lds((glb_void*)(base + offA[t]), (lds_void*)(smem + slot + (8 * (w & 3) + t) * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                int64_t r = (int64_t)rowA0 + 8 * t;
                if (row0 + r > a.n_rows - 1) r = a.n_rows - 1 - row0;
                const uint32_t off = (uint32_t)(r * GF_ROWB) + (offA[t] - (uint32_t)((rowA0 + 8 * t) * GF_ROWB));
                __builtin_amdgcn_global_load_lds((glb_void*)(base + off), (lds_void*)(smem + slot + (8 * (w & 3) + t) * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_b = [&](int kt, int slot) {              // 16 KB, linear (waves 4..7, 4 x 1 KiB each)
        const char* base = uniform_ptr(a.qimg + (size_t)kt * GF_B + (size_t)(4 * (w & 3)) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (uint32_t)(i * 1024 + lane * 16)),
                                             (lds_void*)(smem + slot + (4 * (w & 3) + i) * 1024), 16, 0, 0);
    };
    const bool row_wave = w < 4;                        // wave-uniform role: rows / query pieces
#ifdef MSR_DIAG
    const int dbg = a.dbg;
#else
    constexpr int dbg = 0;
#endif

    float thrv[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) thrv[ni] = EMIT ? a.thr[ni * 16 + li16] : 0.f;
    asm volatile("" :: "v"(thrv[0]), "v"(thrv[1]), "v"(thrv[2]), "v"(thrv[3]), "v"(thrv[4]), "v"(thrv[5]), "v"(thrv[6]),
                 "v"(thrv[7]));                         // (retire these loads before any DMA is in flight)

    f32x4 acc[2][8];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_of = [&](int j) { return a.t_first + j * a.t_stride; };
    int jt = gid;
    int row0 = a.tile_row[tile_of(jt)], row_end = a.tile_row[tile_of(jt) + 1];
    int jn = jt + G < a.t_count ? jt + G : jt;
    int row0n = a.tile_row[tile_of(jn)];
    auto sticks_out = [&](int r0) { return (int64_t)r0 + 256 > a.n_rows; };

    // ---- prologue: rows of steps 0, 1, 2 and the queries of step 0 ----
    if (row_wave) {
        stage_a(row0, 0, a_slot(0), sticks_out(row0));
        stage_a(row0, 1, a_slot(1), sticks_out(row0));
        stage_a(row0, 2, a_slot(2), sticks_out(row0));
    } else {
        stage_b(0, b_slot(0));
    }
    wait_vm0();
    wg_barrier();

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    f32x4 inv4[2];                                      // inverse norms of the lane's 8 rows of the current tile
    inv4[0] = inv4[1] = (f32x4){1.f, 1.f, 1.f, 1.f};
    // K step with row buffer j, query buffer d; issues the rows of step + 3 (waves 0..3) / the queries of step + 1 (4..7)
    auto kstep = [&](auto j_c, auto d_c, int rowN, int ktN, bool clampN, int ktB, int inv_row) {
        constexpr int j = decltype(j_c)::value, d = decltype(d_c)::value;
        if (inv_row >= 0 && !(dbg & 64)) {              // (third step from the end of a tile: this tile's inverse norms)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float* p = a.inv_pad + (size_t)inv_row + w * 32 + mi * 16 + 4 * lg;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(inv4[mi]) : "v"(p) : "memory");
            }
        }
        if (!(dbg & 4)) {
            if (row_wave) stage_a(rowN, ktN, a_slot((j + 3) % GF_NA), clampN);
            else stage_b(ktB, b_slot(d ^ 1));
        }
        // the wave's 32 rows x this step's 32 dimensions: two fragments, split once into f16 hi / lo pieces
        f16x8 ahi[2] = {}, alo[2] = {};
        if (!(dbg & 16)) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const char* p = smem + a_slot(j) + a_base + mi * 2048;
                const f32x4 x0 = *(const f32x4*)(p + foffA[0]), x1 = *(const f32x4*)(p + foffA[1]);
                split_f16(x0, x1, ahi[mi], alo[mi]);
            }
        }
        // ... against all 128 queries, four 16-query blocks at a time
#pragma unroll
        for (int nq4 = 0; nq4 < 2; ++nq4) {
            f16x8 bh[4] = {}, bl[4] = {};
            if (!(dbg & 16)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const char* p = smem + b_slot(d) + (4 * nq4 + i) * 1024 + foffB;
                    bh[i] = *(const f16x8*)p;
                    bl[i] = *(const f16x8*)(p + 8192);
                }
            }
            if (dbg & 8) continue;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mi][4 * nq4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[mi], bh[i], acc[mi][4 * nq4 + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mi][4 * nq4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[mi], bl[i], acc[mi][4 * nq4 + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mi][4 * nq4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[mi], bh[i], acc[mi][4 * nq4 + i], 0, 0, 0);
            }
        }
        // rows: all but the 16 youngest DMAs (the rows of steps + 2 and + 3 stay in flight); queries: everything
        if (row_wave) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else wait_vm0();
        wg_barrier();
    };

    const float NEG_INF = -__builtin_inff();
    for (int it = 0; it < n_mine; ++it) {
        const bool c0 = sticks_out(row0), c1 = sticks_out(row0n);
        // step kt: rows of step kt + 3 (of the next tile once kt + 3 >= 24), queries of step kt + 1
#pragma unroll 1
        for (int k4 = 0; k4 < GF_KT / 4; ++k4) {
            const int kt = 4 * k4;
            auto rn = [&](int s) { return s < GF_KT ? row0 : row0n; };
            auto kn = [&](int s) { return s < GF_KT ? s : s - GF_KT; };
            auto cn = [&](int s) { return s < GF_KT ? c0 : c1; };
            // the plain loads of the inverse norms are issued at the start of step 21: older than the DMAs of steps 21..23,
            // so the ordinary waits of steps 22 and 23 retire them before the epilogue
            kstep(I0{}, I0{}, rn(kt + 3), kn(kt + 3), cn(kt + 3), kn(kt + 1), -1);
            kstep(I1{}, I1{}, rn(kt + 4), kn(kt + 4), cn(kt + 4), kn(kt + 2), kt == GF_KT - 4 ? row0 : -1);
            kstep(I2{}, I0{}, rn(kt + 5), kn(kt + 5), cn(kt + 5), kn(kt + 3), -1);
            kstep(I3{}, I1{}, rn(kt + 6), kn(kt + 6), cn(kt + 6), kn(kt + 4), -1);
        }
        // ---- epilogue: accumulator (mi, ni)[rr] = row 32 w + mi 16 + 4 lg + rr of the tile, query ni 16 + li16 ----
        const int n_valid = row_end - row0;
        int col_e = li16;
        asm volatile("" : "+v"(col_e));                 // (keeps the address arithmetic below inside the tile loop: no spills)
        asm volatile("" : "+v"(inv4[0]), "+v"(inv4[1])); // (read only here, after the waits that retired the loads)
        float cmax[8];
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) cmax[ni] = NEG_INF;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int blk = w * 32 + mi * 16;
            if (blk >= n_valid || (dbg & 32)) continue; // wave-uniform
            const int rb = blk + 4 * lg;
            const bool part = blk + 16 > n_valid;       // wave-uniform
            const f32x4 inv = inv4[mi];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                f32x4 v = acc[mi][ni] * inv;            // cosine = <e, q^> / ||e||
                if (part) {
                    asm volatile("" ::: "memory");      // a real branch (at most one block per wave and tile)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (rb + rr >= n_valid) v[rr] = NEG_INF;
                }
                acc[mi][ni] = v;                        // (kept for the emission pass below)
                cmax[ni] = max2_raw(cmax[ni], max3_raw(max2_raw(v[0], v[1]), v[2], v[3]));
            }
        }
        if (EMIT) {
            // Emission, per 16-query block: only when some lane's maximum over its 8 rows reaches the threshold (about one
            // block in three); every branch is wave-uniform, the position comes from a per-wave scalar counter and a prefix
            // count over the emitting lanes (no atomics: an LDS atomic would wait for every pending LDS-DMA)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                if (__ballot(cmax[ni] >= thrv[ni]) == 0) continue;
                const int q = ni * 16 + col_e;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int rb = w * 32 + mi * 16 + 4 * lg;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float x = acc[mi][ni][rr];
                        const bool hit = x >= thrv[ni];                       // (masked and skipped rows hold -inf / 0 < thr)
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
    }
    wait_vm0();
    if (EMIT && lane == 0) a.wv_count[blockIdx.x * 8 + w] = wave_cnt;
}

// qimg[kt][piece][q][physical chunk c'] (16 B = 8 f16) = piece (hi | lo) of dims 32 kt + 8 c .. + 8 of normalised query q,
// c = c' ^ (((q >> 3) & 1) << 1); queries >= nq are zero.  Same split as the row side (split_f16).
__global__ __launch_bounds__(256) void build_qimg_kernel(const float* __restrict__ qn, int nq, f16x8* __restrict__ qimg) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // (kt, q, c')
    if (i >= GF_KT * 128 * 4) return;
    const int cp = i & 3, q = (i >> 2) & 127, kt = i >> 9;
    const int c = cp ^ (((q >> 3) & 1) << 1);
    f16x8 hi, lo;
    if (q < nq) {
        const float* src = qn + (size_t)q * MSR_DIM + 32 * kt + 8 * c;
        split_f16(*(const f32x4*)src, *(const f32x4*)(src + 4), hi, lo);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; }
    }
    qimg[((size_t)kt * 2 + 0) * 512 + q * 4 + cp] = hi;
    qimg[((size_t)kt * 2 + 1) * 512 + q * 4 + cp] = lo;
}

__global__ __launch_bounds__(256) void pad_inv_kernel(const float* __restrict__ inv, int64_t n, int64_t n_pad, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) out[i] = i < n ? inv[i] : 1.0f;
}

__device__ __forceinline__ bool key_less2(uint64_t ah, uint32_t al, uint64_t bh, uint32_t bl) {
    return ah < bh || (ah == bh && al < bl);
}

// One workgroup per query: entries (row, score) -> per-document maximum with its FIRST arg-max row -> exact order
// (score desc, document asc) -> top-k.  Overflow (entries that did not fit anywhere on the way): out_n = -1 and *gate |= 1.
__global__ __launch_bounds__(1024) void gemm_f32_final_kernel(const int2* __restrict__ pairs, int32_t* __restrict__ pair_n,
                                                               const int32_t* __restrict__ chunk_doc,
                                                               const int32_t* __restrict__ wv_count, int n_waves, int wv_cap,
                                                               const int32_t* __restrict__ flag, int k,
                                                               int32_t* __restrict__ out_doc, float* __restrict__ out_score,
                                                               int32_t* __restrict__ out_chunk, int32_t* __restrict__ out_n,
                                                               int32_t* __restrict__ gate) {
    __shared__ uint64_t khi[GF_PAIR_CAP];
    __shared__ uint32_t klo[GF_PAIR_CAP];
    __shared__ int s_over, s_heads;
    const int q = blockIdx.x, t = threadIdx.x;
    const int raw = pair_n[q];
    // (all threads scan the per-wave counts: one thread walking 2048 words with a data-dependent exit took 0.12 ms)
    int over = raw > GF_PAIR_CAP || flag[q];
    for (int i = t; i < n_waves; i += 1024) over |= wv_count[i] > wv_cap;
    if (t == 0) s_heads = 0;
    s_over = 0;
    __syncthreads();
    if (over) s_over = 1;
    __syncthreads();
    if (s_over) {
        for (int i = t; i < k; i += 1024) {
            out_doc[(size_t)q * k + i] = -1;
            out_score[(size_t)q * k + i] = -__builtin_inff();
            if (out_chunk) out_chunk[(size_t)q * k + i] = -1;
        }
        if (t == 0) { out_n[q] = -1; pair_n[q] = 0; atomicOr(gate, 1); }
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
                    const uint64_t ah = khi[i], bh = khi[p];
                    const uint32_t al = klo[i], bl = klo[p];
                    if (desc ? key_less2(ah, al, bh, bl) : key_less2(bh, bl, ah, al)) {
                        khi[i] = bh; klo[i] = bl; khi[p] = ah; klo[p] = al;
                    }
                }
                __syncthreads();
            }
    };
    // (1) by (document, score, ~row) descending: the first entry of a document's run is its maximum at its lowest row
    for (int i = t; i < P; i += 1024) {
        uint64_t h = 0; uint32_t l = 0;
        if (i < raw) {
            const int2 e = pairs[(size_t)q * GF_PAIR_CAP + i];
            h = ((uint64_t)(uint32_t)(chunk_doc[e.x] + 1) << 32) | msr_ord32(__int_as_float(e.y));   // doc + 1: 0 is the pad key
            l = ~(uint32_t)e.x;
        }
        khi[i] = h; klo[i] = l;
    }
    __syncthreads();
    sort_desc();
    // (2) heads re-keyed by (score, ~document) with the row as payload
    uint64_t mh[GF_PAIR_CAP / 1024];
    uint32_t ml[GF_PAIR_CAP / 1024];
#pragma unroll
    for (int u = 0; u < GF_PAIR_CAP / 1024; ++u) {
        const int i = t + u * 1024;
        uint64_t h = 0; uint32_t l = 0;
        if (i < P && khi[i] != 0 && (i == 0 || (khi[i] >> 32) != (khi[i - 1] >> 32))) {
            h = ((uint64_t)(uint32_t)khi[i] << 32) | (uint32_t)~(uint32_t)((khi[i] >> 32) - 1);
            l = ~klo[i];                                 // the row
        }
        mh[u] = h; ml[u] = l;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GF_PAIR_CAP / 1024; ++u) {
        const int i = t + u * 1024;
        if (i < P) {
            khi[i] = mh[u]; klo[i] = ml[u];
            if (mh[u]) atomicAdd(&s_heads, 1);
        }
    }
    __syncthreads();
    sort_desc();                                        // keys of heads are distinct (the document is part of them)
    const int n_sel = s_heads < k ? s_heads : k;
    for (int i = t; i < k; i += 1024) {
        const bool ok = i < n_sel;
        out_doc[(size_t)q * k + i] = ok ? (int32_t)~(uint32_t)khi[i] : -1;
        out_score[(size_t)q * k + i] = ok ? msr_unord32((uint32_t)(khi[i] >> 32)) : -__builtin_inff();
        if (out_chunk) out_chunk[(size_t)q * k + i] = ok ? (int32_t)klo[i] : -1;
    }
    if (t == 0) { out_n[q] = n_sel; pair_n[q] = 0; }
}

int g_f32_dbg = 0;

hipError_t launch_f32(bool emit, const GemmF32Args& a, int grid, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t err = hipFuncSetAttribute((const void*)gemm_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GF_LDS);
        if (err != hipSuccess) return err;
        err = hipFuncSetAttribute((const void*)gemm_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GF_LDS);
        if (err != hipSuccess) return err;
        attr_done = true;
    }
    if (emit) gemm_f32_kernel<true><<<grid, GF_THREADS, GF_LDS, stream>>>(a);
    else gemm_f32_kernel<false><<<grid, GF_THREADS, GF_LDS, stream>>>(a);
    return hipGetLastError();
}

}  // namespace

void msr_gemm_f32_set_dbg(int v) { g_f32_dbg = v; }

hipError_t msr_pad_inv_norm(const float* inv, int64_t n, int64_t n_pad, float* out, hipStream_t stream) {
    if (n_pad <= 0) return hipSuccess;
    pad_inv_kernel<<<(unsigned)((n_pad + 255) / 256), 256, 0, stream>>>(inv, n, n_pad, out);
    return hipGetLastError();
}

// Exact (f16x2-split arithmetic) top-k of up to 128 queries in one pass over the f32 rows; see the header of this file.
// qn: [nq][768] normalised queries.  out_n[q] = -1 and *gate != 0 when a query overflowed (the caller falls back).
hipError_t msr_gemm_f32_topk(const GemmF32Index& g, const DenseIndex& ix, const float* qn, int nq, int k,
                             const SelScratch& sel, int32_t* out_doc, float* out_score, int32_t* out_chunk,
                             int32_t* out_n, int32_t* gate, hipEvent_t* ev, hipStream_t stream) {
    if (nq <= 0 || nq > 128 || k < 1 || g.n_tiles < 2 * k) return hipErrorInvalidValue;
    hipError_t err;
    build_qimg_kernel<<<(GF_KT * 128 * 4 + 255) / 256, 256, 0, stream>>>(qn, nq, (f16x8*)g.qimg);
    int ss = g.n_tiles / (6 * k);                       // every ss-th tile bounds the k-th score from below: >= 6 k sampled tiles
    ss = ss < 1 ? 1 : (ss > 32 ? 32 : ss);
    const int n_s = (g.n_tiles - ss / 2 + ss - 1) / ss;
    const int grid = g.n_cus;
    GemmF32Args a{};
    a.dbg = g_f32_dbg;
    a.E = (const char*)ix.emb; a.inv_pad = g.inv_pad; a.qimg = (const char*)g.qimg; a.tile_row = g.tile_row;
    a.n_rows = ix.n_chunks; a.tmax_t = g.tmax_t;
    a.t_first = ss / 2; a.t_stride = ss; a.t_count = n_s;
    if (ev && (err = hipEventRecord(ev[0], stream)) != hipSuccess) return err;
    if ((err = launch_f32(false, a, grid, stream)) != hipSuccess) return err;
    if (ev && (err = hipEventRecord(ev[1], stream)) != hipSuccess) return err;
    if ((err = msr_gemm_tmax(g.tmax_t, n_s, 8, 128, g.tmax, g.tmax_stride, stream)) != hipSuccess) return err;
    if ((err = msr_select_topk(32, g.tmax, n_s, g.tmax_stride, nq, k, sel, g.top_doc, g.top_score, g.top_n, stream)) != hipSuccess) return err;
    if ((err = msr_gemm_thr(g.top_score, g.top_n, nq, 128, k, nullptr, g.thr, g.flag, stream)) != hipSuccess) return err;
    a.t_first = 0; a.t_stride = 1; a.t_count = g.n_tiles;
    a.thr = g.thr; a.wvbuf = (int4*)g.wvbuf; a.wv_cap = g.wv_cap; a.wv_count = g.wv_count;
    if (ev && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
    if ((err = launch_f32(true, a, grid, stream)) != hipSuccess) return err;
    if (ev && (err = hipEventRecord(ev[3], stream)) != hipSuccess) return err;
    if ((err = msr_gemm_tmax(g.tmax_t, g.n_tiles, 8, 128, g.tmax, g.tmax_stride, stream)) != hipSuccess) return err;
    if ((err = msr_select_topk(32, g.tmax, g.n_tiles, g.tmax_stride, nq, k, sel, g.top_doc, g.top_score, g.top_n, stream)) != hipSuccess) return err;
    if ((err = msr_gemm_thr(g.top_score, g.top_n, nq, 128, k, nullptr, g.thr2, nullptr, stream)) != hipSuccess) return err;
    if ((err = msr_gemm_bucket(g.wvbuf, g.wv_cap, g.wv_count, grid * 8, g.thr2, g.pairs, g.pair_n, stream)) != hipSuccess) return err;
    gemm_f32_final_kernel<<<nq, 1024, 0, stream>>>((const int2*)g.pairs, g.pair_n, ix.chunk_doc, g.wv_count, grid * 8, g.wv_cap,
                                                   g.flag, k, out_doc, out_score, out_chunk, out_n, gate);
    return hipGetLastError();
}
