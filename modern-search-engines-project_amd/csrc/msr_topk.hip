// K4 -- exact top-k by radix select + final sort (gfx950).
//
// Replaces the reference's full sorts: doc_scores.sort(...)[:top_k] (indexer/bm25_indexer.py:484-485) and
// sort_values(...)[:TOP_K] (reranker/reranker_api.py:372,404).  Order = (score desc, doc index asc), which
// is what Python's stable sort produces on candidates that arrive in ascending doc_id (:445).
//
// Key = (orderable(score), ~index): all keys of a row are distinct, so "the k largest keys" is a unique
// set.  Passes resolve the key 12 bits at a time with an LDS histogram per workgroup; as soon as the
// elements at or above the resolved prefix fit MSR_SEL_CAP they are compacted and one workgroup sorts
// them exactly.  Every pass is a streaming read of the score row: HBM-bound.
#include "msr_common.h"
#include "msr_internal.h"

namespace {

constexpr int SEL_THREADS = 256;
constexpr int SCAN_THREADS = 1024;

template <int SB> struct KeyCfg {
    static constexpr int NS = (SB + 11) / 12;      // digits in the score part
    static constexpr int ND = NS + 3;              // + 12,12,8 bits of ~index
};

template <int SB>
__device__ __forceinline__ void digit_pos(int d, int& part, int& shift, int& width) {
    constexpr int NS = KeyCfg<SB>::NS;
    if (d < NS) {
        part = 0;
        int top = SB - 12 * d;
        width = top < 12 ? top : 12;
        shift = top - width;
    } else {
        part = 1;
        int top = 32 - 12 * (d - NS);
        width = top < 12 ? top : 12;
        shift = top - width;
    }
}

template <typename T> struct ScoreTraits;
template <> struct ScoreTraits<float> {
    static constexpr int SB = 32;
    static __device__ __forceinline__ uint64_t ord(float s) { return msr_ord32(s); }
    static __device__ __forceinline__ float unord(uint64_t u) { return msr_unord32((uint32_t)u); }
    static __device__ __forceinline__ float neg_inf() { return -__builtin_inff(); }
};
template <> struct ScoreTraits<double> {
    static constexpr int SB = 64;
    static __device__ __forceinline__ uint64_t ord(double s) { return msr_ord64(s); }
    static __device__ __forceinline__ double unord(uint64_t u) { return msr_unord64(u); }
    static __device__ __forceinline__ double neg_inf() { return -__builtin_inf(); }
};

__global__ void sel_init_kernel(SelState* st, int k) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int)gridDim.x * (int)blockDim.x) return;
    SelState s;
    s.pref_hi = 0; s.mask_hi = 0; s.pref_lo = 0; s.mask_lo = 0;
    s.k_rem = k; s.n_above = 0; s.done = 0; s.n_sel = 0;
    st[q] = s;
}

template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void sel_hist_kernel(const T* __restrict__ scores, int64_t n,
                                                                int64_t stride, int digit,
                                                                const SelState* __restrict__ st,
                                                                uint32_t* __restrict__ hist) {
    constexpr int SB = ScoreTraits<T>::SB;
    const int q = blockIdx.y;
    const SelState S = st[q];
    if (S.done) return;
    __shared__ uint32_t h[MSR_SEL_BINS];
    for (int b = threadIdx.x; b < MSR_SEL_BINS; b += SEL_THREADS) h[b] = 0;
    __syncthreads();
    int part, shift, width;
    digit_pos<SB>(digit, part, shift, width);
    const uint32_t wmask = (1u << width) - 1u;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    const T* row = scores + (int64_t)q * stride;
    for (int64_t i = lo + threadIdx.x; i < hi; i += SEL_THREADS) {
        const T s = row[i];
        if (!msr_valid(s)) continue;
        const uint64_t khi = ScoreTraits<T>::ord(s);
        const uint32_t klo = ~(uint32_t)i;
        if ((khi & S.mask_hi) != S.pref_hi || (klo & S.mask_lo) != S.pref_lo) continue;
        const uint32_t dg = part == 0 ? (uint32_t)(khi >> shift) & wmask : (klo >> shift) & wmask;
        atomicAdd(&h[dg], 1u);
    }
    __syncthreads();
    uint32_t* gh = hist + (int64_t)q * MSR_SEL_BINS;
    for (int b = threadIdx.x; b < MSR_SEL_BINS; b += SEL_THREADS)
        if (h[b]) atomicAdd(&gh[b], h[b]);
}

template <int SB>
__global__ __launch_bounds__(SCAN_THREADS) void sel_scan_kernel(SelState* __restrict__ st,
                                                                 uint32_t* __restrict__ hist, int digit, int k) {
    const int q = blockIdx.x;
    SelState S = st[q];
    if (S.done) return;                              // nothing was added to hist[q] in this pass
    __shared__ uint32_t h[MSR_SEL_BINS];
    __shared__ uint32_t suf[SCAN_THREADS + 1];
    uint32_t* gh = hist + (int64_t)q * MSR_SEL_BINS;
    const int t = threadIdx.x;
    uint32_t local = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t v = gh[4 * t + j];
        h[4 * t + j] = v;
        gh[4 * t + j] = 0;                           // leave the histogram zeroed for the next pass
        local += v;
    }
    suf[t] = local;
    if (t == 0) suf[SCAN_THREADS] = 0;
    __syncthreads();
    // inclusive suffix sum over threads (Hillis-Steele)
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        uint32_t add = (t + off < SCAN_THREADS) ? suf[t + off] : 0;
        __syncthreads();
        suf[t] += add;
        __syncthreads();
    }
    const uint32_t total = suf[0];
    if (digit == 0 && total <= (uint32_t)S.k_rem) {
        // fewer valid elements than k: everything valid is selected
        if (t == 0) {
            S.done = 1; S.n_sel = (int32_t)total;
            st[q] = S;
        }
        return;
    }
    const uint32_t need = (uint32_t)S.k_rem;
    if (suf[t] >= need && suf[t + 1] < need) {       // exactly one thread
        uint32_t above = suf[t + 1];
        int b = 4 * t + 3;
        for (; b > 4 * t; --b) {
            if (above + h[b] >= need) break;
            above += h[b];
        }
        int part, shift, width;
        digit_pos<SB>(digit, part, shift, width);
        const uint64_t wmask = ((uint64_t)1 << width) - 1;
        if (part == 0) {
            S.pref_hi |= (uint64_t)b << shift; S.mask_hi |= wmask << shift;
        } else {
            S.pref_lo |= (uint32_t)b << shift; S.mask_lo |= (uint32_t)(wmask << shift);
        }
        S.n_above += (int32_t)above;
        S.k_rem -= (int32_t)above;
        S.n_sel = k;
        const uint32_t superset = (uint32_t)S.n_above + h[b];
        if (superset <= MSR_SEL_CAP || digit == KeyCfg<SB>::ND - 1) S.done = 1;
        st[q] = S;
    }
}

template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void sel_compact_kernel(const T* __restrict__ scores, int64_t n,
                                                                   int64_t stride,
                                                                   const SelState* __restrict__ st,
                                                                   uint64_t* __restrict__ cand_hi,
                                                                   uint32_t* __restrict__ cand_lo,
                                                                   int32_t* __restrict__ cand_n) {
    const int q = blockIdx.y;
    const SelState S = st[q];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    const T* row = scores + (int64_t)q * stride;
    for (int64_t i = lo + threadIdx.x; i < hi; i += SEL_THREADS) {
        const T s = row[i];
        if (!msr_valid(s)) continue;
        const uint64_t khi = ScoreTraits<T>::ord(s);
        const uint32_t klo = ~(uint32_t)i;
        const uint64_t mh = khi & S.mask_hi;
        const bool ge = mh > S.pref_hi || (mh == S.pref_hi && (klo & S.mask_lo) >= S.pref_lo);
        if (!ge) continue;
        const int pos = atomicAdd(&cand_n[q], 1);
        if (pos < MSR_SEL_CAP) {
            cand_hi[(int64_t)q * MSR_SEL_CAP + pos] = khi;
            cand_lo[(int64_t)q * MSR_SEL_CAP + pos] = klo;
        }
    }
}

// Bitonic sort of (hi, lo) keys, descending, in LDS.  P is a power of two <= MSR_SEL_CAP * 2.
__device__ __forceinline__ bool key_less(uint64_t ah, uint32_t al, uint64_t bh, uint32_t bl) {
    return ah < bh || (ah == bh && al < bl);
}

template <int THREADS>
__device__ void bitonic_desc(uint64_t* khi, uint32_t* klo, int P) {
    for (int kk = 2; kk <= P; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int idx = threadIdx.x; idx < (P >> 1); idx += THREADS) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int p = i | j;
                const bool desc = (i & kk) == 0;
                const uint64_t ah = khi[i], bh = khi[p];
                const uint32_t al = klo[i], bl = klo[p];
                const bool a_lt_b = key_less(ah, al, bh, bl);
                if (desc ? a_lt_b : !a_lt_b && !(ah == bh && al == bl)) {
                    khi[i] = bh; klo[i] = bl; khi[p] = ah; klo[p] = al;
                }
            }
            __syncthreads();
        }
    }
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void sel_final_kernel(const SelState* __restrict__ st,
                                                                  const uint64_t* __restrict__ cand_hi,
                                                                  const uint32_t* __restrict__ cand_lo,
                                                                  int32_t* __restrict__ cand_n, int k,
                                                                  int32_t* __restrict__ out_doc,
                                                                  T* __restrict__ out_score,
                                                                  int32_t* __restrict__ out_n) {
    __shared__ uint64_t khi[MSR_SEL_CAP];
    __shared__ uint32_t klo[MSR_SEL_CAP];
    const int q = blockIdx.x;
    int n = cand_n[q];
    if (n > MSR_SEL_CAP) n = MSR_SEL_CAP;
    int P = 64;
    while (P < n) P <<= 1;
    for (int i = threadIdx.x; i < P; i += SCAN_THREADS) {
        khi[i] = i < n ? cand_hi[(int64_t)q * MSR_SEL_CAP + i] : 0;
        klo[i] = i < n ? cand_lo[(int64_t)q * MSR_SEL_CAP + i] : 0;
    }
    __syncthreads();
    bitonic_desc<SCAN_THREADS>(khi, klo, P);
    int n_sel = st[q].n_sel;
    if (n_sel > n) n_sel = n;
    for (int i = threadIdx.x; i < k; i += SCAN_THREADS) {
        const bool ok = i < n_sel;
        out_doc[(int64_t)q * k + i] = ok ? (int32_t)~klo[i] : -1;
        out_score[(int64_t)q * k + i] = ok ? ScoreTraits<T>::unord(khi[i]) : ScoreTraits<T>::neg_inf();
    }
    if (threadIdx.x == 0) {
        out_n[q] = n_sel;
        cand_n[q] = 0;                               // invariant: zero between calls
    }
}

template <typename T>
hipError_t select_impl(const T* scores, int64_t n, int64_t stride, int nq, int k, const SelScratch& sc,
                       int32_t* out_doc, T* out_score, int32_t* out_n, hipStream_t stream) {
    constexpr int SB = ScoreTraits<T>::SB;
    if (nq <= 0) return hipSuccess;
    sel_init_kernel<<<nq, 1, 0, stream>>>(sc.state, k);
    int64_t parts = (n + 8191) / 8192;
    const int64_t max_parts = 4096 / nq > 0 ? 4096 / nq : 1;
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    dim3 grid((unsigned)parts, (unsigned)nq);
    for (int d = 0; d < KeyCfg<SB>::ND; ++d) {
        sel_hist_kernel<T><<<grid, SEL_THREADS, 0, stream>>>(scores, n, stride, d, sc.state, sc.hist);
        sel_scan_kernel<SB><<<nq, SCAN_THREADS, 0, stream>>>(sc.state, sc.hist, d, k);
    }
    sel_compact_kernel<T><<<grid, SEL_THREADS, 0, stream>>>(scores, n, stride, sc.state, sc.cand_hi, sc.cand_lo,
                                                             sc.cand_n);
    sel_final_kernel<T><<<nq, SCAN_THREADS, 0, stream>>>(sc.state, sc.cand_hi, sc.cand_lo, sc.cand_n, k, out_doc,
                                                          out_score, out_n);
    return hipGetLastError();
}

// ---- merge of per-shard lists ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void merge_kernel(const int32_t* __restrict__ in_doc,
                                                              const T* __restrict__ in_score,
                                                              const int32_t* __restrict__ in_n, int n_parts,
                                                              int nq, int k, int32_t* __restrict__ out_doc,
                                                              T* __restrict__ out_score,
                                                              int32_t* __restrict__ out_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q = blockIdx.x;
    const int total = n_parts * k;
    int P = 64;
    while (P < total) P <<= 1;
    uint64_t* khi = (uint64_t*)smem;
    uint32_t* klo = (uint32_t*)(smem + (size_t)P * 8);
    __shared__ int n_valid;
    if (threadIdx.x == 0) {
        int s = 0;
        for (int p = 0; p < n_parts; ++p) {
            int c = in_n[p * nq + q];
            s += c < 0 ? 0 : (c > k ? k : c);
        }
        n_valid = s;
    }
    for (int i = threadIdx.x; i < P; i += SCAN_THREADS) {
        uint64_t h = 0; uint32_t l = 0;
        if (i < total) {
            const int p = i / k, r = i % k;
            int c = in_n[p * nq + q];
            if (r < c) {
                const int64_t off = ((int64_t)p * nq + q) * k + r;
                const T s = in_score[off];
                if (msr_valid(s)) { h = ScoreTraits<T>::ord(s); l = ~(uint32_t)in_doc[off]; }
            }
        }
        khi[i] = h; klo[i] = l;
    }
    __syncthreads();
    bitonic_desc<SCAN_THREADS>(khi, klo, P);
    const int n_sel = n_valid < k ? n_valid : k;
    for (int i = threadIdx.x; i < k; i += SCAN_THREADS) {
        const bool ok = i < n_sel && !(khi[i] == 0 && klo[i] == 0);
        out_doc[(int64_t)q * k + i] = ok ? (int32_t)~klo[i] : -1;
        out_score[(int64_t)q * k + i] = ok ? ScoreTraits<T>::unord(khi[i]) : ScoreTraits<T>::neg_inf();
    }
    if (threadIdx.x == 0) out_n[q] = n_sel;
}

}  // namespace

hipError_t msr_select_topk(int score_bits, const void* scores, int64_t n, int64_t stride, int nq, int k,
                           const SelScratch& sc, int32_t* out_doc, void* out_score, int32_t* out_n,
                           hipStream_t stream) {
    if (score_bits == 32)
        return select_impl<float>((const float*)scores, n, stride, nq, k, sc, out_doc, (float*)out_score, out_n, stream);
    return select_impl<double>((const double*)scores, n, stride, nq, k, sc, out_doc, (double*)out_score, out_n, stream);
}

hipError_t msr_merge_lists(int score_bits, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                           int n_parts, int nq, int k, int32_t* out_doc, void* out_score, int32_t* out_n,
                           hipStream_t stream) {
    const int total = n_parts * k;
    int P = 64;
    while (P < total) P <<= 1;
    const size_t lds = (size_t)P * 12;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (score_bits == 32) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)merge_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        merge_kernel<float><<<nq, SCAN_THREADS, lds, stream>>>(in_doc, (const float*)in_score, in_n, n_parts, nq, k,
                                                               out_doc, (float*)out_score, out_n);
    } else {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)merge_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        merge_kernel<double><<<nq, SCAN_THREADS, lds, stream>>>(in_doc, (const double*)in_score, in_n, n_parts, nq, k,
                                                                out_doc, (double*)out_score, out_n);
    }
    return hipGetLastError();
}
