// Query encoder, the matrix products (include/msretr_encoder.h msr_enc_linear).
//
// y[T][N] = x[T][K] . w[N][K]^T (+ resid[T][N]) for the four projections of a ModernBERT layer
// (N, K) in {(2304, 768), (768, 768), (2304, 768), (768, 1152)}: the reference's encoder
// (reranker/reranker_api.py:137-139,355; transformers ModernBertModel) runs them as torch.nn.Linear.  A query has
// 3..128 tokens, a batch of queries a few thousand, so the products are SKINNY: the 20 MB of weights of a layer are
// the traffic, the tokens fit in L2.  Layout of the work, the same as the dense scan's (msr_dense.hip):
//   * v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation): 16 weight rows are the A operand, 16 tokens the B
//     operand; lane (n = lane & 15, g = lane >> 4) loads w[row n][16t + 4g .. +3] and x[token n][16t + 4g .. +3] --
//     one 16-byte load feeds four MFMAs, the k index of an MFMA is only a label, no data moves between lanes.
//   * a workgroup owns RB x 16 weight rows and NTB x 16 tokens; its 4 waves split K in four slices (a weight element is
//     read from HBM once; token tiles beyond the first find it in L2), partial sums meet in LDS in a fixed order
//     (wave 0..3: the result does not depend on scheduling), the residual is added there and lane (n, g) writes
//     y[token n][16 rb + 4g .. +3] as one 16-byte store.
//   * N / (16 RB) x ceil(T / (16 NTB)) workgroups: 48..144 for a single query (latency-bound: 22 x 4 of these sit in
//     one hipGraph with the other encoder kernels), thousands for a batch.
#include "../../include/msretr.h"
#include "../../include/msretr_encoder.h"
#include "msr_common.h"
#include "msr_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NTB, int RB, int PF>
__global__ __launch_bounds__(256) void enc_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* resid, float* y,   // (resid may alias y)
                                                         int n_tok, int n_out, int n_in) {
    __shared__ f32x4 red[4][RB][NTB][64];                     // partial sums of the four K slices
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * (16 * RB);
    const int tok0 = blockIdx.y * (16 * NTB);
    const int kslice = n_in >> 2;                             // per wave; a multiple of 32 (host checks n_in % 128 == 0)
    const int kb = wave * kslice + 4 * g;

    const float* wp[RB];
    const float* xp[NTB];
#pragma unroll
    for (int r = 0; r < RB; ++r) wp[r] = w + (int64_t)(row0 + 16 * r + n) * n_in + kb;
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        int tok = tok0 + 16 * tb + n;
        tok = tok < n_tok ? tok : n_tok - 1;                  // rows past the end: clamped load, never stored
        xp[tb] = x + (int64_t)tok * n_in + kb;
    }

    f32x4 acc[RB][NTB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[r][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // k-steps of 32: lane (n, g) loads floats 32 t + 8 g .. + 7 of its rows (two 16-byte loads), so the four lanes g of a row
    // consume one whole 128-byte line per step and no line is visited twice (with 16-float steps every line was fetched in
    // two steps, and between them the workgroups of a CU had pushed it out of the 32 KB vector cache: three workgroups per CU
    // ran 1.4x SLOWER than one).  The operands of PF steps are in flight ahead of the MFMAs (a ring of PF register sets,
    // statically indexed: the loop body is unrolled PF times).  Which k an MFMA multiplies is only a label: both operands
    // use the same one.
    const int steps = kslice >> 5;
    const int kb2 = wave * kslice + 8 * g - kb;                // (the pointers were set up for 4 g: move them to 8 g)
    f32x4 a[PF][RB][2], b[PF][NTB][2];
    auto load = [&](int p, int t) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            a[p][r][0] = *reinterpret_cast<const f32x4*>(wp[r] + kb2 + 32 * t);
            a[p][r][1] = *reinterpret_cast<const f32x4*>(wp[r] + kb2 + 32 * t + 4);
        }
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            b[p][tb][0] = *reinterpret_cast<const f32x4*>(xp[tb] + kb2 + 32 * t);
            b[p][tb][1] = *reinterpret_cast<const f32x4*>(xp[tb] + kb2 + 32 * t + 4);
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < steps) load(p, p);
    for (int t0 = 0; t0 < steps; t0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int t = t0 + p;
            if (t < steps) {                                  // (uniform)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < RB; ++r)
#pragma unroll
                            for (int tb = 0; tb < NTB; ++tb)
                                acc[r][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][r][hh][c], b[p][tb][hh][c], acc[r][tb], 0, 0, 0);
                if (t + PF < steps) load(p, t + PF);
            }
        }
    }

#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) red[wave][r][tb][lane] = acc[r][tb];
    __syncthreads();
    // tile (r, tb) is finished by wave (r NTB + tb) & 3; acc[r][tb][i] = y[token tok0 + 16 tb + n][row0 + 16 r + 4g + i]
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            if (((r * NTB + tb) & 3) != wave) continue;
            const int tok = tok0 + 16 * tb + n;
            f32x4 s = red[0][r][tb][lane];
            s = s + red[1][r][tb][lane];
            s = s + red[2][r][tb][lane];
            s = s + red[3][r][tb][lane];
            if (tok < n_tok) {
                const int64_t o = (int64_t)tok * n_out + row0 + 16 * r + 4 * g;
                if (resid) s = s + *reinterpret_cast<const f32x4*>(resid + o);
                *reinterpret_cast<f32x4*>(y + o) = s;
            }
        }
}

int g_enc_shape = 0;                                          // (only the diagnostic build can set these two)
int g_enc_pad_lds = 0;

template <int NTB, int RB, int PF>
void launch(const float* x, const float* w, const float* resid, float* y, int n_tok, int n_out, int n_in,
            hipStream_t stream, int pad_lds = 0) {
    const dim3 grid((unsigned)(n_out / (16 * RB)), (unsigned)((n_tok + 16 * NTB - 1) / (16 * NTB)));
    enc_linear_kernel<NTB, RB, PF><<<grid, 256, pad_lds, stream>>>(x, w, resid, y, n_tok, n_out, n_in);
}

}  // namespace

extern "C" int msr_enc_linear(const float* x, const float* w, const float* resid, float* y, int32_t n_tok,
                              int32_t n_out, int32_t n_in, void* stream) {
    if (!x || !w || !y || n_tok < 0 || n_out < 32 || n_out % 32 || n_in < 128 || n_in % 128)
        return msr_fail_global(MSR_ERR_INVALID,
                               "msr_enc_linear: bad argument (n_out=%d must be a multiple of 32, n_in=%d of 128)", n_out,
                               n_in);
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)resid) & 15)
        return msr_fail_global(MSR_ERR_INVALID, "msr_enc_linear: pointers must be 16-byte aligned");
    if (n_tok == 0) return MSR_OK;
    hipStream_t s = (hipStream_t)stream;
    // token blocks per workgroup: the smallest that covers a single query's tokens (fewer idle MFMAs; the whole K slice
    // of a wave is in flight at once), 64-token tiles beyond that
    if (n_tok <= 16) launch<1, 1, 3>(x, w, resid, y, n_tok, n_out, n_in, s);
    else if (n_tok <= 32) launch<2, 1, 2>(x, w, resid, y, n_tok, n_out, n_in, s);
    else if (n_tok <= 128) launch<4, 1, 2>(x, w, resid, y, n_tok, n_out, n_in, s);
    else {
        // Batches (measured at 1024 tokens, profiles/r03_enc_linear_shapes.md): 64 tokens x 48 weight rows per workgroup
        // (7 operand loads per 48 MFMAs), and ONE workgroup per CU -- each has a wave on every SIMD and the kernel is bound by
        // the f32 matrix rate, so co-resident workgroups only take turns at the matrix pipes and evict each other's lines:
        // 2304 x 768 ran 54 us with three (two) workgroups per CU, 43 us with one (the library GEMM: 40 us).  The occupancy
        // is set with unused dynamic LDS.  Output widths that are not a multiple of 48 take 32 rows per workgroup.
        const int shape = g_enc_shape ? g_enc_shape : (n_out % 48 == 0 ? 1 : 2);
        const int pad = g_enc_shape ? g_enc_pad_lds : 40 * 1024;
        switch (shape) {
            case 1: launch<4, 3, 2>(x, w, resid, y, n_tok, n_out, n_in, s, pad); break;
            case 2: launch<4, 2, 2>(x, w, resid, y, n_tok, n_out, n_in, s, pad); break;
            case 3: launch<2, 3, 2>(x, w, resid, y, n_tok, n_out, n_in, s, pad); break;
            case 4: launch<2, 2, 2>(x, w, resid, y, n_tok, n_out, n_in, s, pad); break;
            default: launch<4, 4, 1>(x, w, resid, y, n_tok, n_out, n_in, s, pad); break;
        }
    }
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? MSR_OK : msr_fail_global(MSR_ERR_HIP, "msr_enc_linear: %s", hipGetErrorString(err));
}

#ifdef MSR_DIAG
extern "C" void msr_enc_linear_force_shape(int code) { g_enc_shape = code & 15; g_enc_pad_lds = (code >> 4) * 1024; }   // timing experiments:
// low 4 bits 1 .. 5 = candidate shape (0 = choose), the rest = KB of unused LDS per workgroup
#endif
