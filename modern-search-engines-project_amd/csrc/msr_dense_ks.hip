// K2 + K3, wide sweeps -- one pass over the embedding matrix for up to 64 queries (f32 rows) or 128 (bf16 rows).
//
// Same job as dense_scan_v2_kernel (msr_dense.hip; reference: reranker/reranker_api.py:273-287 cosines and the
// per-document arg-max :370 at full-corpus scale), but decomposed so that the query side needs NO LDS: the wave
// streaming kernel keeps the whole query image in LDS (96 KB at 32 queries), which caps a sweep at 32 queries
// and makes every extra query cost another full read of E.  Here the K dimension is split over the 8 waves of
// a workgroup: wave w owns the 96 dimensions [96 w, 96 w + 96) of every row and holds the matching slice of ALL
// queries' fragments in registers for the whole kernel (the B operand of every MFMA is a loop-invariant VGPR).
//
//   per 16-row group ("unit") and wave:
//     6 x 16 B loads per lane (its 384 B slice of 16 rows; 64 B of a row per instruction; prefetched NBUF-1 units
//     ahead in a register ring), f32 -> f16 hi/lo split, 3 k-steps x 3 products x QB query blocks MFMAs into QB
//     accumulators, the 16 x Q partial sums go to LDS (one of two buffers), ONE workgroup barrier;
//     then every wave adds the 8 partial tiles for its eighth of the (row, query) pairs in a fixed order, scales
//     by the row's inverse norm and folds the value into the document's slot of a ring of per-document maxima in
//     LDS with ds_max_f32 -- no wave carries sequential per-document state, so this work is spread evenly.
//   Documents are finished in order: when the first row of document b + 32 has been passed, the 32 maxima of the
//   aligned block [b, b + 32) are written (whole 128 B lines of the score rows, all waves share the block) and
//   their ring slots reset to -inf, which is also what a chunk-less document reports.
//
// The ring holds 128 documents; it never wraps onto live slots if any 32 consecutive rows (on 16-row group
// boundaries) span fewer than 96 documents, which the engine checks when the chunks are bound (DenseIndex.wide_ok;
// otherwise the narrow kernel runs).  max_chunks > 0 (a per-document row limit) also stays on the narrow kernel.
//
// The workgroups are persistent, one per CU, each with its own span of rows cut at a document boundary, so there
// is no traffic between workgroups.  Bytes per launch are those of the narrow kernel (E once, 15.36 GB at 5 M
// rows) for twice the queries; the LDS traffic is the partial tiles only (64 KB per 48 KB of rows).
#include <stdlib.h>

#include "msr_common.h"
#include "msr_internal.h"
#include "msr_frag.h"

namespace {

template <int QB, int MODE, int RING_DOCS = MSR_WIDE_RING, int NWAVES = 8> struct KsCfg {
    static constexpr int WAVES = NWAVES;                             // K-split ways; waves 0..7 also reduce and write
    static constexpr int THREADS = WAVES * 64;
    static constexpr int KS = MSR_DIM / 32;                          // MFMA k-steps per row
    static constexpr int KT = KS / WAVES;                            // k-steps per wave (3, or 2 with 12 waves)
    static constexpr int PIECES = MODE == MODE_BF16 ? 1 : 2;         // 16 B fragment pieces per 32 dimensions (f16 modes: hi, lo;
                                                                     // exact f32: two blocks of 16 dimensions)
    static constexpr int NLU = MODE == MODE_BF16 ? KT : 2 * KT;      // 16 B loads per lane and unit
    static constexpr int ROW16 = MODE == MODE_BF16 ? MSR_DIM * 2 / 16 : MSR_DIM * 4 / 16;
    static constexpr int NQ = 16 * QB;                               // padded query count
    static constexpr int RING = RING_DOCS;                           // documents in the ring of maxima
    static constexpr int SWZ = (NQ < 64 ? NQ : 64) - 1;              // column swizzle mask of the ring
    static constexpr int TILE = QB * 64 * 4;                         // floats of one 16-row x NQ tile
    static constexpr int PER = QB / 2;                               // tile floats per lane in the reduction
    static constexpr size_t p_bytes = (size_t)2 * WAVES * TILE * 4;  // partial tiles of the waves, two buffers
    static constexpr size_t r_bytes = (size_t)RING * NQ * 4;
    static constexpr size_t total = p_bytes + r_bytes;
    static_assert(QB == 2 || QB == 4 || QB == 8, "the reduction hands QB / 2 floats (one tile column piece) to a lane");
    static_assert(KS % WAVES == 0, "k-steps split evenly over the waves");
    static_assert(MODE == MODE_BF16 || MODE == MODE_F16X2 || MODE == MODE_PRE || MODE == MODE_F32, "supported products");
};

template <int QB, int MODE, int NBUF, int PIPE, int RING_DOCS, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void dense_ksplit_kernel(DenseIndex ix, const void* __restrict__ emb,
                                                           const f32x4* __restrict__ qimg, int nq,
                                                           float* __restrict__ docscore, int dbg) {
    // blockIdx.y = slice of 16 QB queries of the call (one slice in every launch but the gated fallback of the streaming
    // pass, msr_dense_scan_slices: there a slice runs only if its gate word is up)
    using L = KsCfg<QB, MODE, RING_DOCS, NWAVES>;
    constexpr int KT = L::KT, NLU = L::NLU, PIECES = L::PIECES, NQ = L::NQ, RING = L::RING, PER = L::PER;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* P = (f32x4*)smem;                                     // [2][8 waves][QB][64 lanes]
    float* R = (float*)(smem + L::p_bytes);                      // [RING][NQ], column q of slot s at q ^ (s & SWZ)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // PIPE: 0 = every wave reduces a unit right behind its barrier; 1 = every wave reduces it after the NEXT unit's
    // MFMAs (LDS reads in flight meanwhile); 2 = waves 4..7 do, waves 0..3 do not: the two waves of a SIMD then run their
    // MFMA phase and their LDS/VALU phase in opposite order and overlap each other
    const bool red = w < 8;                                      // waves 0..7 reduce and write; 12-wave instances: 8..11 only multiply
    const bool pw = red && (PIPE == 1 || (PIPE == 2 && w >= 4));
    const int s = blockIdx.x;
    if (s >= ix.n_spans) return;                                 // workgroup-uniform
    const int sl = blockIdx.y;
    if (ix.gate && ix.gate[sl] == 0) return;                     // a fallback slice that is not needed (msr_engine.hip)
    qimg += (size_t)sl * (QB * L::KS * PIECES * 64);
    docscore += (int64_t)sl * NQ * ix.score_stride;
    nq = nq - sl * NQ < NQ ? nq - sl * NQ : NQ;
    const int64_t C = ix.n_chunks;
    const float NEG_INF = -__builtin_inff();
    const int d0 = ix.span_doc[s], d1 = ix.span_doc[s + 1];
    const int64_t c0 = ix.doc_off[d0], c1 = ix.doc_off[d1];
    const int dbase = d0 & ~31;                                  // ring slot of document d: (d - dbase) & (RING - 1)

    // this wave's slice of every query fragment: loop-invariant registers
    f32x4 B[QB][KT][PIECES];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int tt = 0; tt < KT; ++tt)
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc)
                B[qb][tt][pc] = qimg[((size_t)(qb * L::KS + KT * w + tt) * PIECES + pc) * 64 + lane];   // (same index
                    // arithmetic for the f32 image: its 16-dimension block 2 (KT w + tt) + pc)

    for (int i = tid; i < RING * NQ; i += L::THREADS) R[i] = NEG_INF;

    // ---- finished documents: blocks of 32, in order; every wave runs the same scalar bookkeeping ----
    int next_b = dbase;                                          // first document of the next block to write
    auto block_end_row = [&](int b) -> int64_t { return ix.doc_off[b + 32 < d1 ? b + 32 : d1]; };
    int64_t next_end = block_end_row(next_b);                    // once rows below this index are in, the block is complete
    auto flush_block = [&](int b) {
        if (w >= 8) return;                                      // (12-wave instances: waves 8..11 only multiply)
        // wave w writes queries [2 QB w, 2 QB (w + 1)): 2 queries x 32 documents per instruction, 128 B per query
        const int dd = lane & 31, qq = lane >> 5;
        const int slot = ((b - dbase) & (RING - 1)) + dd;
        const int d = b + dd;
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const int q = 2 * QB * w + 2 * i + qq;
            float* cell = &R[slot * NQ + (q ^ (slot & L::SWZ))];
            const float v = *cell;
            *cell = NEG_INF;
            if (q < nq && d >= d0 && d < d1 && !(dbg & 1)) docscore[(int64_t)q * ix.score_stride + d] = v;
        }
    };
    // Inside the loop at most two blocks become complete per unit (the engine admits the kernel only if a 16-row group
    // spans <= 32 documents), so two plain ifs do: a loop with stores in it would make the compiler wait for ALL
    // outstanding loads, prefetches included, at the next use of a row register.
    auto flush_step = [&](int64_t rows_done) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (next_b < d1 && next_end <= rows_done) {
                flush_block(next_b);
                next_b += 32;
                if (next_b < d1) next_end = block_end_row(next_b);
            }
    };
    auto flush_done = [&](int64_t rows_done) {                   // rows [c0, rows_done) have been folded in
        while (next_b < d1 && next_end <= rows_done) {
            flush_block(next_b);
            next_b += 32;
            if (next_b < d1) next_end = block_end_row(next_b);
        }
    };

    // Tile float f = ((qb * 4 + g) * 16 + n) * 4 + rr is row 4 g + rr of query 16 qb + n (D layout of the 16x16 MFMA:
    // lane (n, g) holds rows 4 g .. 4 g + 3).  Lane l of wave w reduces floats f0 .. f0 + PER - 1.
    const int f0 = (w * 64 + lane) * PER;
    const int red_q = 16 * (f0 >> 8) + ((f0 >> 2) & 15);
    const int red_r0 = 4 * ((f0 >> 6) & 3) + (f0 & 3);
    const int64_t g0 = c0 >> 4, g1 = c1 > c0 ? (c1 + 15) >> 4 : g0;
    auto clampg = [&](int64_t g) { return g < g1 ? g : g1 - 1; };
    auto row_ptr = [&](int64_t grp) -> const f32x4* {
        int64_t r = clampg(grp) * 16 + li;
        if (r > C - 1) r = C - 1;
        return (const f32x4*)emb + (size_t)r * L::ROW16 + NLU * 4 * w + lg;
    };
    struct Meta { int d[PER]; float inv[PER]; };                 // document and inverse norm of the lane's reduce rows
    auto load_meta = [&](int64_t grp, Meta& mt) {                // PER consecutive {doc, inv} pairs: 8 PER bytes, aligned
        const int2* src = (const int2*)ix.row_meta + clampg(grp) * 16 + red_r0;   // (padded: no clamp at the last row)
        if constexpr (PER == 1) {
            const int2 x = src[0];
            mt.d[0] = x.x; mt.inv[0] = __int_as_float(x.y);
        } else {
#pragma unroll
            for (int j = 0; j < PER; j += 2) {
                const int4 x = *(const int4*)(src + j);
                mt.d[j] = x.x; mt.inv[j] = __int_as_float(x.y);
                mt.d[j + 1] = x.z; mt.inv[j + 1] = __int_as_float(x.w);
            }
        }
    };
    // The reduction of a unit comes in two halves so that (PIPE) the LDS reads can be in flight during the next
    // unit's MFMAs: red_load issues them, red_finish adds in a fixed order (wave 0 .. 7: the result does not depend
    // on timing) and folds the cosines into the ring.
    typedef float redvec __attribute__((ext_vector_type(PER == 1 ? 1 : PER)));
    redvec raw[NWAVES];
    auto red_load = [&](int buf) {
#pragma unroll
        for (int w8 = 0; w8 < NWAVES; ++w8)
            raw[w8] = *(const redvec*)((const float*)P + (size_t)(buf * NWAVES + w8) * L::TILE + f0);
    };
    auto red_finish = [&](int64_t grp, const Meta& mt) {
        redvec sum = raw[0];
#pragma unroll
        for (int w8 = 1; w8 < NWAVES; ++w8) sum += raw[w8];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int64_t row = grp * 16 + red_r0 + j;
            if (row >= c0 && row < c1) {                         // not a row of a neighbouring span / past the end
                const int slot = (mt.d[j] - dbase) & (RING - 1);
                __hip_atomic_fetch_max(&R[slot * NQ + (red_q ^ (slot & L::SWZ))], sum[j] * mt.inv[j], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };

    Meta mt_prev = {};
    __syncthreads();                                             // ring initialised
    // A span (or a shard) may BEGIN with chunk-less documents: the bind-time check (wide_ok) only looks at the
    // documents between consecutive rows, so write those leading blocks now -- otherwise next_b would lag behind the
    // documents being folded (the loop retires at most two blocks per unit) and ring slots would alias.
    flush_done(c0);
    if (g1 > g0) {
        f32x4 A[NBUF][NLU];
        Meta mtr[NBUF] = {};
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b) {
            const f32x4* p = row_ptr(g0 + b);
#pragma unroll
            for (int j = 0; j < NLU; ++j) A[b][j] = p[j * 4];
            load_meta(g0 + b, mtr[b]);
        }
        __builtin_amdgcn_sched_barrier(0);

        for (int64_t base = g0; base < g1; base += NBUF) {
#pragma unroll
            for (int ph = 0; ph < NBUF; ++ph) {
                const int64_t grp = base + ph;
                if (grp >= g1) break;                            // workgroup-uniform
                const int nx = (ph + NBUF - 1) % NBUF;           // ring slot of the unit being prefetched
                {
                    const f32x4* pn = row_ptr(grp + NBUF - 1);
#pragma unroll
                    for (int j = 0; j < NLU; ++j) A[nx][j] = pn[j * 4];
                    load_meta(grp + NBUF - 1, mtr[nx]);
                }
                if (pw && grp > g0) red_load((int)((grp - 1 - g0) & 1));        // unit u - 1: complete since barrier u - 1
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) acc[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE == MODE_F32) {
                    // exact f32: four v_mfma_f32_16x16x4_f32 per 16 B load and query block (the k index is only a label,
                    // see msr_dense.hip); bit-for-bit a k-ordered fmaf chain per K slice, slices added in wave order
#pragma unroll
                    for (int j = 0; j < NLU; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb)
                                acc[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ph][j][c], B[qb][j / 2][j % 2][c], acc[qb], 0, 0, 0);
                } else if constexpr (MODE == MODE_F16X2 || MODE == MODE_PRE) {
#pragma unroll
                    for (int tt = 0; tt < KT; ++tt) {
                        f16x8 ahi, alo;
                        if constexpr (MODE == MODE_PRE) {       // the image holds the pieces where the floats would be
                            ahi = __builtin_bit_cast(f16x8, A[ph][2 * tt]);
                            alo = __builtin_bit_cast(f16x8, A[ph][2 * tt + 1]);
                        } else {
                            split_f16(A[ph][2 * tt], A[ph][2 * tt + 1], ahi, alo);
                        }
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb)
                                acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    pc == 0 ? alo : ahi, __builtin_bit_cast(f16x8, pc == 1 ? B[qb][tt][PIECES - 1] : B[qb][tt][0]),
                                    acc[qb], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int tt = 0; tt < KT; ++tt)
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[ph][tt]),
                                                                              __builtin_bit_cast(bf16x8, B[qb][tt][0]),
                                                                              acc[qb], 0, 0, 0);
                }
                const int buf = (int)((grp - g0) & 1);
                if (pw) {
                    // finish unit u - 1 (its partial tiles were read before the MFMAs above)
                    if (grp > g0) {
                        red_finish(grp - 1, mt_prev);
                        flush_step((grp - 1) * 16);
                    }
                    mt_prev = mtr[ph];
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) P[((buf * NWAVES + w) * QB + qb) * 64 + lane] = acc[qb];
                // Barrier u: unit u's partial tiles are complete and so are the ring updates of unit u - 2.  P[u & 1] is
                // written again for unit u + 2, after barrier u + 1, which a wave only reaches once it has read its part
                // of unit u.  Between barriers u and u + 1 every wave folds unit u into the ring -- right here, or
                // (pipelined waves) after the next unit's MFMAs -- and writes the blocks that are complete through
                // unit u - 1; the ring updates other waves issue meanwhile belong to later documents, i.e. other slots.
                __syncthreads();
                if (!pw && red) {
                    red_load(buf);
                    red_finish(grp, mtr[ph]);
                    flush_step(grp * 16);
                }
            }
        }
        if (pw) {
            red_load((int)((g1 - 1 - g0) & 1));
            red_finish(g1 - 1, mt_prev);
        }
        __syncthreads();                                         // the last unit's ring updates
    }
    flush_done((int64_t)1 << 62);                                // what is left, chunk-less tail included
}

__global__ __launch_bounds__(256) void pack_row_meta_kernel(const int32_t* __restrict__ chunk_doc,
                                                            const float* __restrict__ inv_norm, int64_t n,
                                                            int2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n + 16) return;
    const int64_t r = i < n ? i : n - 1;
    out[i] = make_int2(chunk_doc[r], __float_as_int(inv_norm[r]));
}

// dst f32x4 slot (row, 2T, g) <- f16 hi pieces, slot (row, 2T + 1, g) <- lo pieces of the 8 floats mode 2 takes from those
// two slots (T = k-step of 32 dimensions, g = lane >> 4)
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ src, int64_t n_rows,
                                                       f32x4* __restrict__ dst) {
    const int64_t n = n_rows * (MSR_DIM / 32) * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int g = (int)(i & 3);
        const int T = (int)((i >> 2) % (MSR_DIM / 32));
        const int64_t r = i / (4 * (MSR_DIM / 32));
        const f32x4* row = (const f32x4*)(src + r * MSR_DIM);
        f16x8 hi, lo;
        split_f16(row[4 * (2 * T) + g], row[4 * (2 * T + 1) + g], hi, lo);
        f32x4* out = dst + r * (MSR_DIM / 4);
        out[4 * (2 * T) + g] = __builtin_bit_cast(f32x4, hi);
        out[4 * (2 * T + 1) + g] = __builtin_bit_cast(f32x4, lo);
    }
}

// Diagnostic build only (-DMSR_DIAG): MSR_SCAN_DEBUG bit 0 drops the score-row stores (timing experiments).
int scan_debug_flags() {
#ifdef MSR_DIAG
    static const int v = [] { const char* e = getenv("MSR_SCAN_DEBUG"); return e ? atoi(e) : 0; }();
    return v;
#else
    return 0;
#endif
}

// Diagnostic build only: MSR_KS_PIPE selects A/B instances of the K-split kernel; the product build returns `dflt`.
int ks_pipe_knob(int dflt) {
#ifdef MSR_DIAG
    static const char* v = getenv("MSR_KS_PIPE");
    return v ? atoi(v) : dflt;
#else
    return dflt;
#endif
}

template <int QB, int MODE, int NBUF, int PIPE = 1, int RING_DOCS = MSR_WIDE_RING, int NWAVES = 8>
hipError_t launch_ksplit(const DenseIndex& ix, const void* emb, const float* qn, int nq, float* docscore,
                         hipStream_t stream, int n_slices = 1, void* qimg_slices = nullptr) {
    using L = KsCfg<QB, MODE, RING_DOCS, NWAVES>;
    static_assert(L::total <= 160 * 1024, "LDS budget");
    hipError_t err = hipFuncSetAttribute((const void*)dense_ksplit_kernel<QB, MODE, NBUF, PIPE, RING_DOCS, NWAVES>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::total);
    if (err != hipSuccess) return err;
    void* qimg = qimg_slices ? qimg_slices : ix.qimg;           // (ix.qimg holds ONE slice's image)
    err = msr_build_qimage(MODE == MODE_PRE ? MODE_F16X2 : MODE, qn, QB * n_slices, qimg, stream);
    if (err != hipSuccess) return err;
    dense_ksplit_kernel<QB, MODE, NBUF, PIPE, RING_DOCS, NWAVES><<<dim3((unsigned)ix.n_spans, (unsigned)n_slices), L::THREADS,
                                                                 L::total, stream>>>(
        ix, emb, (const f32x4*)qimg, nq, docscore, scan_debug_flags());
    return hipGetLastError();
}

}  // namespace

hipError_t msr_pack_row_meta(const int32_t* chunk_doc, const float* inv_norm, int64_t n, void* row_meta,
                             hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    pack_row_meta_kernel<<<(unsigned)((n + 16 + 255) / 256), 256, 0, stream>>>(chunk_doc, inv_norm, n, (int2*)row_meta);
    return hipGetLastError();
}

hipError_t msr_presplit_rows(const float* src, int64_t n_rows, void* dst, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    presplit_kernel<<<16384, 256, 0, stream>>>(src, n_rows, (f32x4*)dst);
    return hipGetLastError();
}

// f32 rows, f16-split products, up to 64 queries per sweep (row-major layout, ix.wide_ok).
hipError_t msr_dense_scan_wide(const DenseIndex& ix, const float* qn, int nq, float* docscore, hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    if (nq > 64 || ix.layout != 0 || !ix.wide_ok || !ix.row_meta) return hipErrorInvalidValue;
    // 64 queries: 12-way K split (2 k-steps per wave, 167 VGPRs: three waves per SIMD), 2-3 % faster than the 8-way split
    // (profile r01_m); 32 queries: 8-way.  Diagnostic build only: MSR_KS_PIPE=8 -> 8 waves; 2 -> 8 waves with the
    // reduction pipelined in waves 4..7 only (see PIPE in the kernel; no gain measured).
    const int pipe = ks_pipe_knob(0);
    if (ix.variant == 15 && ix.emb_presplit) {              // A/B variant: pre-split copy of the rows (+4 bytes per value of HBM)
        if (nq <= 32) return launch_ksplit<2, MODE_PRE, 3, 0>(ix, ix.emb_presplit, qn, nq, docscore, stream);
        return launch_ksplit<4, MODE_PRE, 2, 0, MSR_WIDE_RING, 12>(ix, ix.emb_presplit, qn, nq, docscore, stream);
    }
    if (nq <= 32) return launch_ksplit<2, MODE_F16X2, 3, 0>(ix, ix.emb, qn, nq, docscore, stream);   // (8 waves: 12 gain nothing here)
#ifdef MSR_DIAG
    if (pipe == 2) return launch_ksplit<4, MODE_F16X2, 3, 2>(ix, ix.emb, qn, nq, docscore, stream);
    if (pipe == 8) return launch_ksplit<4, MODE_F16X2, 3, 0>(ix, ix.emb, qn, nq, docscore, stream);
#endif
    (void)pipe;
    return launch_ksplit<4, MODE_F16X2, 2, 0, MSR_WIDE_RING, 12>(ix, ix.emb, qn, nq, docscore, stream);
}

// The gated fallback behind the streaming pass: ceil(nq / 64) slices of 64 queries in ONE launch (grid.y = slice), slice s
// gated on ix.gate[s]; qn holds the normalised queries padded with zero rows to a multiple of 64, qimg_slices room for
// msr_ksplit_slice_image_bytes() per slice, docscore 64 rows per slice.  f16-split products (the 64-query instance).
size_t msr_ksplit_slice_image_bytes() { return (size_t)4 * KsCfg<4, MODE_F16X2, MSR_WIDE_RING, 12>::KS * 2 * 64 * 16; }
hipError_t msr_dense_scan_slices(const DenseIndex& ix, const float* qn, int nq, float* docscore, void* qimg_slices,
                                 hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    if (ix.layout != 0 || !ix.wide_ok || !ix.row_meta || !ix.gate || !qimg_slices || ix.variant != 14) return hipErrorInvalidValue;
    return launch_ksplit<4, MODE_F16X2, 2, 0, MSR_WIDE_RING, 12>(ix, ix.emb, qn, nq, docscore, stream, (nq + 63) / 64, qimg_slices);
}

// f32 rows, exact f32 products (v_mfma_f32_16x16x4_f32), up to 64 queries per sweep: matrix-core bound above 32 queries.
hipError_t msr_dense_scan_wide_exact(const DenseIndex& ix, const float* qn, int nq, float* docscore, hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    if (nq > 64 || ix.layout != 0 || !ix.wide_ok || !ix.row_meta) return hipErrorInvalidValue;
    if (nq <= 32) return launch_ksplit<2, MODE_F32, 3, 0>(ix, ix.emb, qn, nq, docscore, stream);
    return launch_ksplit<4, MODE_F32, 2, 0, MSR_WIDE_RING, 12>(ix, ix.emb, qn, nq, docscore, stream);
}

// bf16 rows (candidate generator of the batched path): up to 64 queries per sweep, or up to 128 with a ring of 64
// documents when the corpus allows it (ix.wide_ok64; 128 queries' partial tiles leave 32 KB of LDS for the ring).
hipError_t msr_dense_scan_bf16_wide(const DenseIndex& ix, const float* qn, int nq, float* docscore,
                                    hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    if (nq > 128 || !ix.emb_bf16 || !ix.wide_ok || !ix.row_meta || (nq > 64 && !ix.wide_ok64)) return hipErrorInvalidValue;
    const int pipe = ks_pipe_knob(1);
    if (nq > 64) {
        // (PIPE would need 32 more registers than a wave has here)
        return launch_ksplit<8, MODE_BF16, 3, 0, 64>(ix, ix.emb_bf16, qn, nq, docscore, stream);
    }
#ifdef MSR_DIAG
    if (!pipe) return launch_ksplit<4, MODE_BF16, 4, 0>(ix, ix.emb_bf16, qn, nq, docscore, stream);
#endif
    (void)pipe;
    return launch_ksplit<4, MODE_BF16, 4, 1>(ix, ix.emb_bf16, qn, nq, docscore, stream);
}
