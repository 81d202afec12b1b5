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
// Kernel shape: persistent workgroup per CU, 8 waves as 4 (rows) x 2 (queries), tile 256 rows x 128 queries, K step 32:
//   LDS  3 x 32 KB of rows (256 x 128 B, LDS-DMA, two steps of flight) + 2 x 16 KB of query pieces (hi | lo, one step)
//        + 2 x 1 KB of inverse row norms (per tile); 16 B chunks of a row XOR-swizzled so the 32 B fragment reads
//        (two ds_read_b128) are conflict-free;
//   per step and wave: 8 + 8 fragment reads, 4 splits (VALU), 48 MFMAs, vmcnt(4), ONE barrier.
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
constexpr int GF_INV = 3 * GF_A + 2 * GF_B;     // offset of the inverse-norm buffers (2 x 1 KB)
constexpr int GF_LDS = GF_INV + 2048;
constexpr int GF_PAIR_CAP = 4096;

struct GemmF32Args {
    const char* E;             // f32 [n_rows][768] (caller's matrix: NOT padded, the last tile clamps its row index)
    const float* inv_pad;      // [n_rows + 512] inverse norms (engine-owned padded copy)
    const char* qimg;          // [24 K steps][hi | lo][128 queries][64 B] f16 pieces, chunk-swizzled (build_qimg_kernel)
    const int32_t* tile_row;   // [n_tiles + 1]
    int64_t n_rows;
    int t_first, t_stride, t_count;
    float* tmax_t;             // [t_count][4 wave rows][128]
    const float* thr;          // [128] emit threshold (+inf: never)                                   -- emit pass only
    int4* wvbuf; int wv_cap; int32_t* wv_count;   // per-wave emission buffers {row, query, score bits, tile}   -- emit pass only
};

template <bool EMIT>
__global__ __launch_bounds__(GF_THREADS) void gemm_f32_kernel(GemmF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
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
    // DMA of a row buffer: instruction t (0..3) of wave w fills rows 8 (4 w + t) .. +8: lane -> (row R, physical chunk lane & 7),
    // which holds logical chunk c = (lane & 7) ^ f(R & 15), f(r) = ((r >> 1) & 3) << 1 | (r >> 3)
    int rowA[4];
    uint32_t chkA[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        rowA[t] = 8 * (4 * w + t) + (lane >> 3);
        const int f = (((lane >> 4) & 3) << 1) | (t & 1);        // R & 15 = 8 (t & 1) + (lane >> 3)
        chkA[t] = (uint32_t)(((lane & 7) ^ f) * 16);
    }
    // fragment reads: row li16 of a 16-row block, logical chunks 2 lg and 2 lg + 1 (32 B = 8 floats)
    const int fr = (((li16 >> 1) & 3) << 1) | ((li16 >> 3) & 1);
    uint32_t foffA[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) foffA[h] = (uint32_t)(li16 * 128 + (((2 * lg + h) ^ fr) << 4));
    const uint32_t foffB = (uint32_t)(li16 * 64 + ((lg ^ (((li16 >> 3) & 1) << 1)) << 4));
    const uint32_t a_base = (uint32_t)(wr * 64 * 128), b_base = (uint32_t)(wc * 64 * 64);

    auto a_slot = [](int j) { return j * GF_A; };
    auto b_slot = [](int d) { return 3 * GF_A + d * GF_B; };
    // rows of the tile that starts at `row0`, K step kt -> row buffer; `clamp`: the tile may stick out of the matrix
    auto stage_a = [&](int row0, int kt, int slot, bool clamp) {
        const char* base = a.E + (size_t)kt * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int64_t r = (int64_t)row0 + rowA[t];
            if (clamp && r > a.n_rows - 1) r = a.n_rows - 1;
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r * GF_ROWB + chkA[t]),
                                             (lds_void*)(smem + slot + (4 * w + t) * 1024), 16, 0, 0);
        }
    };
    auto stage_b = [&](int kt, int slot) {              // 16 KB, linear
        const char* src = a.qimg + (size_t)kt * GF_B + lane * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + (2 * w + i) * 1024), (lds_void*)(smem + slot + (2 * w + i) * 1024), 16, 0, 0);
    };
    auto stage_inv = [&](int row0, int par) {           // 256 inverse norms: waves 0..3, one 256 B piece each
        if (w < 4)
            __builtin_amdgcn_global_load_lds((glb_void*)(a.inv_pad + (size_t)row0 + w * 64 + lane),
                                             (lds_void*)(smem + GF_INV + par * 1024 + w * 256), 4, 0, 0);
    };

    float thrv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) thrv[ni] = EMIT ? a.thr[wc * 64 + ni * 16 + li16] : 0.f;
    asm volatile("" :: "v"(thrv[0]), "v"(thrv[1]), "v"(thrv[2]), "v"(thrv[3]));    // (retire these loads before any DMA is in flight)

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_of = [&](int j) { return a.t_first + j * a.t_stride; };
    int jt = gid;
    int row0 = a.tile_row[tile_of(jt)], row_end = a.tile_row[tile_of(jt) + 1];
    int jn = jt + G < a.t_count ? jt + G : jt;
    int row0n = a.tile_row[tile_of(jn)];
    auto sticks_out = [&](int r0) { return (int64_t)r0 + 256 > a.n_rows; };

    // ---- prologue: rows of steps 0 and 1, queries of step 0, inverse norms of the first tile ----
    stage_inv(row0, 0);
    stage_b(0, b_slot(0));
    stage_a(row0, 0, a_slot(0), sticks_out(row0));
    stage_a(row0, 1, a_slot(1), sticks_out(row0));
    wait_vm0();
    wg_barrier();

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // K step with row buffer j, query buffer d; issues the queries of the next step and the rows of the step after it
    auto kstep = [&](auto j_c, auto d_c, int rowN, int ktN, bool clampN, int ktB, int inv_row, int inv_par) {
        constexpr int j = decltype(j_c)::value, d = decltype(d_c)::value;
        if (inv_row >= 0) stage_inv(inv_row, inv_par);  // (second step of a tile: the NEXT tile's inverse norms)
        stage_b(ktB, b_slot(d ^ 1));
        stage_a(rowN, ktN, a_slot((j + 2) % 3), clampN);
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const char* p = smem + b_slot(d) + b_base + ni * 1024 + foffB;
            bh[ni] = *(const f16x8*)p;
            bl[ni] = *(const f16x8*)(p + 8192);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const char* p = smem + a_slot(j) + a_base + mi * 2048;
            const f32x4 x0 = *(const f32x4*)(p + foffA[0]), x1 = *(const f32x4*)(p + foffA[1]);
            f16x8 ahi, alo;
            split_f16(x0, x1, ahi, alo);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh[ni], acc[mi][ni], 0, 0, 0);
        }
        wait_vm4();                                     // everything but the 4 youngest DMAs (the rows of step + 2)
        wg_barrier();
    };

    const float NEG_INF = -__builtin_inff();
    for (int it = 0; it < n_mine; ++it) {
        const bool c0 = sticks_out(row0), c1 = sticks_out(row0n);
        // step kt: rows of step kt + 2 (of the next tile once kt + 2 >= 24), queries of step kt + 1
#pragma unroll 1
        for (int k6 = 0; k6 < GF_KT / 6; ++k6) {
            const int kt = 6 * k6;
            auto rn = [&](int s) { return s < GF_KT ? row0 : row0n; };
            auto kn = [&](int s) { return s < GF_KT ? s : s - GF_KT; };
            auto cn = [&](int s) { return s < GF_KT ? c0 : c1; };
            // (the next tile's inverse norms go into the buffer the PREVIOUS epilogue read: not before every wave has passed
            // the barrier of this tile's first step, i.e. has finished that epilogue)
            kstep(I0{}, I0{}, rn(kt + 2), kn(kt + 2), cn(kt + 2), kn(kt + 1), -1, 0);
            kstep(I1{}, I1{}, rn(kt + 3), kn(kt + 3), cn(kt + 3), kn(kt + 2), kt == 0 ? row0n : -1, (it + 1) & 1);
            kstep(I2{}, I0{}, rn(kt + 4), kn(kt + 4), cn(kt + 4), kn(kt + 3), -1, 0);
            kstep(I0{}, I1{}, rn(kt + 5), kn(kt + 5), cn(kt + 5), kn(kt + 4), -1, 0);
            kstep(I1{}, I0{}, rn(kt + 6), kn(kt + 6), cn(kt + 6), kn(kt + 5), -1, 0);
            kstep(I2{}, I1{}, rn(kt + 7), kn(kt + 7), cn(kt + 7), kn(kt + 6), -1, 0);
        }
        // ---- epilogue: accumulator (mi, ni)[rr] = row wr 64 + mi 16 + 4 lg + rr of the tile, query wc 64 + ni 16 + li16 ----
        const int n_valid = row_end - row0;
        int col_e = li16;
        asm volatile("" : "+v"(col_e));                 // (keeps the address arithmetic below inside the tile loop: no spills)
        const float* invs = (const float*)(smem + GF_INV + (it & 1) * 1024);
        float cmax[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int blk = wr * 64 + mi * 16;
            if (blk >= n_valid) continue;               // wave-uniform
            const int rb = blk + 4 * lg;
            const bool part = blk + 16 > n_valid;       // wave-uniform
            const f32x4 inv = *(const f32x4*)(invs + rb);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[mi][ni] * inv;            // cosine = <e, q^> / ||e||
                if (part) {
                    asm volatile("" ::: "memory");      // a real branch (at most one block per wave and tile)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (rb + rr >= n_valid) v[rr] = NEG_INF;
                }
                const float m = max3_raw(max2_raw(v[0], v[1]), v[2], v[3]);
                cmax[ni] = max2_raw(cmax[ni], m);
                if (EMIT && __ballot(m >= thrv[ni]) != 0) {          // wave-uniform branches only (see msr_gemm.hip)
                    const int q = wc * 64 + ni * 16 + col_e;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const bool hit = v[rr] >= thrv[ni];
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
        for (int ni = 0; ni < 4; ++ni) {
            float m = cmax[ni];
            auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = max2_raw(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
            auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = max2_raw(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
            if (lg == 0) a.tmax_t[((size_t)jt * 4 + wr) * 128 + wc * 64 + ni * 16 + col_e] = m;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
    if (t == 0) {
        int over = raw > GF_PAIR_CAP || flag[q];
        for (int i = 0; i < n_waves && !over; ++i) over = wv_count[i] > wv_cap;
        s_over = over;
        s_heads = 0;
    }
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
    int ss = g.n_tiles / (8 * k);
    ss = ss < 1 ? 1 : (ss > 16 ? 16 : ss);
    const int n_s = (g.n_tiles - ss / 2 + ss - 1) / ss;
    const int grid = g.n_cus;
    GemmF32Args a{};
    a.E = (const char*)ix.emb; a.inv_pad = g.inv_pad; a.qimg = (const char*)g.qimg; a.tile_row = g.tile_row;
    a.n_rows = ix.n_chunks; a.tmax_t = g.tmax_t;
    a.t_first = ss / 2; a.t_stride = ss; a.t_count = n_s;
    if (ev && (err = hipEventRecord(ev[0], stream)) != hipSuccess) return err;
    if ((err = launch_f32(false, a, grid, stream)) != hipSuccess) return err;
    if (ev && (err = hipEventRecord(ev[1], stream)) != hipSuccess) return err;
    if ((err = msr_gemm_tmax(g.tmax_t, n_s, 4, 128, g.tmax, g.tmax_stride, stream)) != hipSuccess) return err;
    if ((err = msr_select_topk(32, g.tmax, n_s, g.tmax_stride, nq, k, sel, g.top_doc, g.top_score, g.top_n, stream)) != hipSuccess) return err;
    if ((err = msr_gemm_thr(g.top_score, g.top_n, nq, 128, k, nullptr, g.thr, g.flag, stream)) != hipSuccess) return err;
    a.t_first = 0; a.t_stride = 1; a.t_count = g.n_tiles;
    a.thr = g.thr; a.wvbuf = (int4*)g.wvbuf; a.wv_cap = g.wv_cap; a.wv_count = g.wv_count;
    if (ev && (err = hipEventRecord(ev[2], stream)) != hipSuccess) return err;
    if ((err = launch_f32(true, a, grid, stream)) != hipSuccess) return err;
    if (ev && (err = hipEventRecord(ev[3], stream)) != hipSuccess) return err;
    if ((err = msr_gemm_tmax(g.tmax_t, g.n_tiles, 4, 128, g.tmax, g.tmax_stride, stream)) != hipSuccess) return err;
    if ((err = msr_select_topk(32, g.tmax, g.n_tiles, g.tmax_stride, nq, k, sel, g.top_doc, g.top_score, g.top_n, stream)) != hipSuccess) return err;
    if ((err = msr_gemm_thr(g.top_score, g.top_n, nq, 128, k, nullptr, g.thr2, nullptr, stream)) != hipSuccess) return err;
    if ((err = msr_gemm_bucket(g.wvbuf, g.wv_cap, g.wv_count, grid * 8, g.thr2, g.pairs, g.pair_n, stream)) != hipSuccess) return err;
    gemm_f32_final_kernel<<<nq, 1024, 0, stream>>>((const int2*)g.pairs, g.pair_n, ix.chunk_doc, g.wv_count, grid * 8, g.wv_cap,
                                                   g.flag, k, out_doc, out_score, out_chunk, out_n, gate);
    return hipGetLastError();
}
