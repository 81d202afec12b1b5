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

    // k-steps of 16; the operands of PF steps are in flight ahead of the MFMAs (a ring of PF register sets, statically
    // indexed: the loop body is unrolled PF times)
    const int steps = kslice >> 4;
    f32x4 a[PF][RB], b[PF][NTB];
    auto load = [&](int p, int t) {
#pragma unroll
        for (int r = 0; r < RB; ++r) a[p][r] = *reinterpret_cast<const f32x4*>(wp[r] + 16 * t);
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) b[p][tb] = *reinterpret_cast<const f32x4*>(xp[tb] + 16 * t);
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
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int tb = 0; tb < NTB; ++tb)
                            acc[r][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][r][c], b[p][tb][c], acc[r][tb], 0, 0, 0);
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

template <int NTB, int RB, int PF>
void launch(const float* x, const float* w, const float* resid, float* y, int n_tok, int n_out, int n_in,
            hipStream_t stream) {
    const dim3 grid((unsigned)(n_out / (16 * RB)), (unsigned)((n_tok + 16 * NTB - 1) / (16 * NTB)));
    enc_linear_kernel<NTB, RB, PF><<<grid, 256, 0, stream>>>(x, w, resid, y, n_tok, n_out, n_in);
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
    // of a wave is in flight at once), 64-token tiles beyond that; batches (> 128 tokens) take 32 weight rows per
    // workgroup, which halves the L2 traffic of x
    if (n_tok <= 16) launch<1, 1, 6>(x, w, resid, y, n_tok, n_out, n_in, s);
    else if (n_tok <= 32) launch<2, 1, 4>(x, w, resid, y, n_tok, n_out, n_in, s);
    else if (n_tok <= 128) launch<4, 1, 3>(x, w, resid, y, n_tok, n_out, n_in, s);
    else launch<4, 2, 3>(x, w, resid, y, n_tok, n_out, n_in, s);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? MSR_OK : msr_fail_global(MSR_ERR_HIP, "msr_enc_linear: %s", hipGetErrorString(err));
}
