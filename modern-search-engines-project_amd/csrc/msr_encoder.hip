// Query encoder, the parts between the matrix products (include/msretr_encoder.h).
//
// The reference encodes the query with a sentence-transformers ModernBERT-base bi-encoder
// (reranker/reranker_api.py:137-139,355).  Its GEMMs are plain library GEMMs (hipBLASLt via torch); the kernels here
// are the rest of the forward pass for the shapes a query has -- a few tokens to a few thousand tokens in total,
// sequences of at most 128 tokens: LayerNorm (with the embedding lookup fused in), rotary embedding + attention per
// (sequence, head), GeGLU, masked mean pooling.  All float32.
#include <math.h>

#include "../../include/msretr.h"
#include "../../include/msretr_encoder.h"
#include "msr_common.h"
#include "msr_internal.h"

namespace {

constexpr int HEAD_DIM = 64;
constexpr int MAX_SEQ = 128;

// One wave per row: y = (x - mean) / sqrt(var + eps) * w, biased variance (torch.nn.LayerNorm), no bias.
// dim = 64 * PER; lane l holds elements l, l + 64, ...
template <int PER>
__global__ __launch_bounds__(256) void enc_layernorm_kernel(const float* __restrict__ x, const int32_t* __restrict__ ids,
                                                            const float* __restrict__ table,
                                                            const float* __restrict__ w, float* __restrict__ y,
                                                            int64_t n_rows, float eps) {
    constexpr int dim = 64 * PER;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const float* src = x ? x + row * dim : table + (int64_t)ids[row] * dim;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = src[lane + 64 * j];
        s += v[j];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const float d = v[j] - mean;
        q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)dim + eps);
#pragma unroll
    for (int j = 0; j < PER; ++j) y[row * dim + lane + 64 * j] = (v[j] - mean) * rstd * w[lane + 64 * j];
}

// One workgroup per (sequence, head), one thread per query token (<= 128).  K (rotary embedding applied) and V of the
// head sit in LDS; every thread rotates its own q row and folds the keys in one pass with a running maximum and a
// running denominator (the usual streaming form of softmax(q.k / 8) . V).
__global__ __launch_bounds__(MAX_SEQ) void enc_attention_kernel(const float* __restrict__ qkv,
                                                                const int32_t* __restrict__ seq_off, int n_heads,
                                                                const float* __restrict__ inv_freq, int window,
                                                                float* __restrict__ out) {
    __shared__ float Ks[MAX_SEQ][HEAD_DIM];
    __shared__ float Vs[MAX_SEQ][HEAD_DIM];
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x;
    const int t0 = seq_off[b], S = seq_off[b + 1] - t0;
    const int stride = 3 * n_heads * HEAD_DIM;
    float q[HEAD_DIM];
#pragma unroll
    for (int j = 0; j < HEAD_DIM; ++j) q[j] = 0.f;
    if (t < S) {
        const float* base = qkv + (int64_t)(t0 + t) * stride + h * HEAD_DIM;
        const float* kp = base + n_heads * HEAD_DIM;
        const float* vp = base + 2 * n_heads * HEAD_DIM;
        // rotate-half: x'_j = x_j cos_j - x_{j+32} sin_j (j < 32), x'_j = x_j cos_{j-32} + x_{j-32} sin_{j-32} (j >= 32),
        // angle_j = position * inv_freq[j]
#pragma unroll
        for (int j = 0; j < HEAD_DIM / 2; ++j) {
            const float ang = (float)t * inv_freq[j];
            const float c = cosf(ang), sn = sinf(ang);
            const float q1 = base[j], q2 = base[j + 32], k1 = kp[j], k2 = kp[j + 32];
            q[j] = q1 * c - q2 * sn;
            q[j + 32] = q2 * c + q1 * sn;
            Ks[t][j] = k1 * c - k2 * sn;
            Ks[t][j + 32] = k2 * c + k1 * sn;
            Vs[t][j] = vp[j];
            Vs[t][j + 32] = vp[j + 32];
        }
    }
    __syncthreads();
    if (t >= S) return;
    float mx = -INFINITY, den = 0.f;
    float acc[HEAD_DIM];
#pragma unroll
    for (int j = 0; j < HEAD_DIM; ++j) acc[j] = 0.f;
    for (int k = 0; k < S; ++k) {
        const int dist = k > t ? k - t : t - k;
        if (window > 0 && dist > window) continue;               // outside the local window: weight 0
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < HEAD_DIM; ++j) d += q[j] * Ks[k][j];
        const float sc = d * 0.125f;                             // head_dim ** -0.5
        const float m_new = fmaxf(mx, sc);
        const float scale = expf(mx - m_new);                    // first key: exp(-inf) = 0
        const float p = expf(sc - m_new);
        den = den * scale + p;
#pragma unroll
        for (int j = 0; j < HEAD_DIM; ++j) acc[j] = acc[j] * scale + p * Vs[k][j];
        mx = m_new;
    }
    float* o = out + (int64_t)(t0 + t) * (n_heads * HEAD_DIM) + h * HEAD_DIM;
#pragma unroll
    for (int j = 0; j < HEAD_DIM; ++j) o[j] = acc[j] / den;
}

// The same for SHORT sequences (at most SMAX tokens each: a batch of queries has 3 .. 20 tokens per sequence): one WAVE per
// (sequence, head) -- the kernel above gives such a pair a 128-thread workgroup and 64 KB of LDS, of which a query uses 8
// threads.  Q, K (both rotated) and V of the pair sit in the wave's own slice of LDS (rows padded to 65 floats: a column
// read hits 64 different banks); lane p computes the scores of (query token, key token) pair p, lane t the softmax of row t,
// lane j output feature j of every token.  No workgroup barrier: the four waves of a workgroup are independent, a wave's LDS
// operations execute in program order.
template <int SMAX>
__global__ __launch_bounds__(256) void enc_attention_short_kernel(const float* __restrict__ qkv,
                                                                  const int32_t* __restrict__ seq_off, int n_seq, int n_heads,
                                                                  const float* __restrict__ inv_freq, int window,
                                                                  float* __restrict__ out) {
    constexpr int LD = HEAD_DIM + 1;
    __shared__ float Qs[4][SMAX][LD], Ks[4][SMAX][LD], Vs[4][SMAX][LD];
    __shared__ float Ps[4][SMAX][SMAX + 1];                      // exp(score - row max); column SMAX: the row's denominator
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    const int b = pair / n_heads, h = pair - b * n_heads;
    if (b >= n_seq) return;                                      // (wave-uniform)
    const int t0 = seq_off[b], S = seq_off[b + 1] - t0;
    if (S <= 0) return;
    const int stride = 3 * n_heads * HEAD_DIM;
    float* o = out + (int64_t)t0 * (n_heads * HEAD_DIM) + h * HEAD_DIM + lane;
    if (S > SMAX) {                                              // the caller's length bound was wrong: fail loudly, touch nothing else
        for (int t = 0; t < S; ++t) o[(int64_t)t * (n_heads * HEAD_DIM)] = __builtin_nanf("");
        return;
    }
    float (*Q)[LD] = Qs[wave], (*K)[LD] = Ks[wave], (*V)[LD] = Vs[wave];
    float (*P)[SMAX + 1] = Ps[wave];
    // rotate-half: x'_j = x_j cos_j - x_{j+32} sin_j (j < 32), x'_j = x_j cos_{j-32} + x_{j-32} sin_{j-32} (j >= 32),
    // angle_j = position * inv_freq[j]; lane j owns feature j
    {
        const float f = inv_freq[lane & 31];
        const float sgn = lane < 32 ? -1.f : 1.f;
        for (int t = 0; t < S; ++t) {
            const float* base = qkv + (int64_t)(t0 + t) * stride + h * HEAD_DIM;
            const float ang = (float)t * f;
            const float c = cosf(ang), sn = sinf(ang);
            const float q1 = base[lane], q2 = base[lane ^ 32];
            const float k1 = base[n_heads * HEAD_DIM + lane], k2 = base[n_heads * HEAD_DIM + (lane ^ 32)];
            Q[t][lane] = q1 * c + sgn * (q2 * sn);
            K[t][lane] = k1 * c + sgn * (k2 * sn);
            V[t][lane] = base[2 * n_heads * HEAD_DIM + lane];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // scores of the S x S (query token, key token) pairs, 64 pairs at a time
    for (int p = lane; p < S * S; p += 64) {
        const int t = p / S, k = p - t * S;
        const int dist = k > t ? k - t : t - k;
        float sc = -INFINITY;                                    // outside the local window: weight 0
        if (!(window > 0 && dist > window)) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < HEAD_DIM; ++j) d += Q[t][j] * K[k][j];
            sc = d * 0.125f;                                     // head_dim ** -0.5
        }
        P[t][k] = sc;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < S) {                                              // softmax of row `lane` (its own key is never masked: max is finite)
        float mx = -INFINITY, den = 0.f;
        for (int k = 0; k < S; ++k) mx = fmaxf(mx, P[lane][k]);
        for (int k = 0; k < S; ++k) {
            const float e = expf(P[lane][k] - mx);
            P[lane][k] = e;
            den += e;
        }
        P[lane][SMAX] = den;
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < S; ++t) {                                // lane j: feature j of token t
        float a = 0.f;
        for (int k = 0; k < S; ++k) a += P[t][k] * V[k][lane];
        o[(int64_t)t * (n_heads * HEAD_DIM)] = a / P[t][SMAX];
    }
}

__global__ __launch_bounds__(256) void enc_geglu_kernel(const float* __restrict__ u, float* __restrict__ y,
                                                        int64_t n_rows, int half) {
    const int64_t n = n_rows * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / half;
        const int j = (int)(i - m * half);
        const float a = u[m * 2 * half + j], g = u[m * 2 * half + half + j];
        y[i] = 0.5f * a * (1.0f + erff(a * 0.70710678118654752440f)) * g;      // torch GELU, approximate="none"
    }
}

// One workgroup per sequence; thread j owns features j, j + 256, ...
__global__ __launch_bounds__(256) void enc_mean_pool_kernel(const float* __restrict__ h,
                                                            const int32_t* __restrict__ seq_off, int dim,
                                                            int normalize, float* __restrict__ out) {
    __shared__ float red[256];
    const int b = blockIdx.x, t0 = seq_off[b], S = seq_off[b + 1] - t0;
    float ss = 0.f;
    for (int j = threadIdx.x; j < dim; j += 256) {
        float s = 0.f;
        for (int t = 0; t < S; ++t) s += h[(int64_t)(t0 + t) * dim + j];
        const float m = S > 0 ? s / (float)S : 0.f;
        out[(int64_t)b * dim + j] = m;
        ss += m * m;
    }
    if (!normalize) return;
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float nrm = fmaxf(sqrtf(red[0]), 1e-12f);              // torch.nn.functional.normalize eps
    for (int j = threadIdx.x; j < dim; j += 256) out[(int64_t)b * dim + j] /= nrm;
}

}  // namespace

extern "C" int msr_enc_layernorm(const float* x, const int32_t* ids, const float* table, const float* w, float* y,
                                 int64_t n_rows, int32_t dim, float eps, void* stream) {
    if ((!x && (!ids || !table)) || !w || !y || n_rows < 0 || (dim != 768 && dim != 1024))
        return msr_fail_global(MSR_ERR_INVALID, "msr_enc_layernorm: bad argument (dim=%d; 768 or 1024)", dim);
    if (n_rows == 0) return MSR_OK;
    const unsigned grid = (unsigned)((n_rows + 3) / 4);
    if (dim == 768) enc_layernorm_kernel<12><<<grid, 256, 0, (hipStream_t)stream>>>(x, ids, table, w, y, n_rows, eps);
    else enc_layernorm_kernel<16><<<grid, 256, 0, (hipStream_t)stream>>>(x, ids, table, w, y, n_rows, eps);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? MSR_OK : msr_fail_global(MSR_ERR_HIP, "msr_enc_layernorm: %s", hipGetErrorString(err));
}

extern "C" int msr_enc_attention(const float* qkv, const int32_t* seq_off, int32_t n_seq, int32_t n_heads,
                                 const float* inv_freq, int32_t window, int32_t max_len, float* out, void* stream) {
    if (!qkv || !seq_off || !inv_freq || !out || n_seq < 0 || n_heads < 1 || n_heads > 64 || max_len > MAX_SEQ)
        return msr_fail_global(MSR_ERR_INVALID, "msr_enc_attention: bad argument");
    if (n_seq == 0) return MSR_OK;
    // (sequence lengths are on the device; the caller guarantees <= max_len (<= 128) tokens per sequence, encoder.py checks it)
    const unsigned pairs = (unsigned)n_seq * (unsigned)n_heads;
    hipStream_t st = (hipStream_t)stream;
    if (max_len >= 1 && max_len <= 8)
        enc_attention_short_kernel<8><<<(pairs + 3) / 4, 256, 0, st>>>(qkv, seq_off, n_seq, n_heads, inv_freq, window, out);
    else if (max_len >= 1 && max_len <= 16)
        enc_attention_short_kernel<16><<<(pairs + 3) / 4, 256, 0, st>>>(qkv, seq_off, n_seq, n_heads, inv_freq, window, out);
    else if (max_len >= 1 && max_len <= 32)
        enc_attention_short_kernel<32><<<(pairs + 3) / 4, 256, 0, st>>>(qkv, seq_off, n_seq, n_heads, inv_freq, window, out);
    else
        enc_attention_kernel<<<dim3((unsigned)n_seq, (unsigned)n_heads), MAX_SEQ, 0, st>>>(qkv, seq_off, n_heads, inv_freq, window, out);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? MSR_OK : msr_fail_global(MSR_ERR_HIP, "msr_enc_attention: %s", hipGetErrorString(err));
}

extern "C" int msr_enc_geglu(const float* u, float* y, int64_t n_rows, int32_t half, void* stream) {
    if (!u || !y || n_rows < 0 || half < 1) return msr_fail_global(MSR_ERR_INVALID, "msr_enc_geglu: bad argument");
    if (n_rows == 0) return MSR_OK;
    int64_t blocks = (n_rows * half + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    enc_geglu_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(u, y, n_rows, half);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? MSR_OK : msr_fail_global(MSR_ERR_HIP, "msr_enc_geglu: %s", hipGetErrorString(err));
}

extern "C" int msr_enc_mean_pool(const float* h, const int32_t* seq_off, int32_t n_seq, int32_t dim, int32_t normalize,
                                 float* out, void* stream) {
    if (!h || !seq_off || !out || n_seq < 0 || dim < 1) return msr_fail_global(MSR_ERR_INVALID, "msr_enc_mean_pool: bad argument");
    if (n_seq == 0) return MSR_OK;
    enc_mean_pool_kernel<<<(unsigned)n_seq, 256, 0, (hipStream_t)stream>>>(h, seq_off, dim, normalize, out);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? MSR_OK : msr_fail_global(MSR_ERR_HIP, "msr_enc_mean_pool: %s", hipGetErrorString(err));
}
