// K4 -- exact top-k by radix select + final sort (gfx950).
//
// Replaces the reference's full sorts: doc_scores.sort(...)[:top_k] (indexer/bm25_indexer.py:484-485) and
// sort_values(...)[:TOP_K] (reranker/reranker_api.py:372,404).  Order = (score desc, doc index asc), which
// is what Python's stable sort produces on candidates that arrive in ascending doc_id (:445).
//
// Key = (orderable(score), ~index): all keys of a row are distinct, so "the k largest keys" is a unique
// set.  Two streaming passes resolve the top 24 key bits with LDS histograms (12 bits each) -- or, for BM25 lists whose
// scores are bounded from the query alone, ONE pass over a window of 4096 consecutive 20-bit prefixes (msr_internal.h) --;
// the elements at or above the resolved prefix (<= MSR_SEL_CAP in every non-degenerate case) are compacted and one
// workgroup resolves further digits on them until they fit the next power of two above k, then sorts them exactly.  Tie
// groups too large for that are finished digit by digit inside the final kernel.  Every pass is a streaming read of the
// score row.  Merges of sorted lists (the shards' lists after an all-gather): merge_rank_kernel (counting, over the lists'
// prefixes above a cut) with merge_kernel (a merge tree) behind it for what it declines.
#include "msr_common.h"
#include "msr_internal.h"
#include "msr_sort.h"

#ifdef MSR_DIAG
// diagnostic build: phase timestamps of sel_final_kernel's workgroup 0 (s_memtime ticks), read by msr_debug_sel_final
__device__ unsigned long long msr_dbg_ts[16];
extern "C" int msr_debug_sel_final(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(msr_dbg_ts), sizeof(unsigned long long) * 16);
}
#define DBG_TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) msr_dbg_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DBG_TS(i) do { } while (0)
#endif

namespace {

constexpr int SEL_THREADS = 256;
constexpr int SCAN_THREADS = 1024;
constexpr int WIN_SHIFT = 44;                      // the window pass bins 20-bit prefixes of a 64-bit key (msr_internal.h)

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

// Block-wide step of the radix select on an LDS histogram h[MSR_SEL_BINS] (SCAN_THREADS threads, all of
// them must call it).  Finds the bin that holds the S.k_rem-th largest element among the elements counted in
// h, appends that digit to the resolved prefix and updates the counters.  `suf` is LDS scratch of
// SCAN_THREADS + 1 words, `S_sh` an LDS copy of the state that every thread reads back.
// digit = -1: the window pass (h counts the 20-bit prefixes win_base .. win_base + 4095, both ends clamped).
template <int SB>
__device__ void select_step(const uint32_t* h, uint32_t* suf, SelState* S_sh, int digit, int k, int* superset_out = nullptr,
                            uint64_t win_base = 0) {
    const int t = threadIdx.x;
    uint32_t local = h[4 * t] + h[4 * t + 1] + h[4 * t + 2] + h[4 * t + 3];
    suf[t] = local;
    if (t == 0) suf[SCAN_THREADS] = 0;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {           // inclusive suffix sum over threads
        const uint32_t add = (t + off < SCAN_THREADS) ? suf[t + off] : 0;
        __syncthreads();
        suf[t] += add;
        __syncthreads();
    }
    const SelState S = *S_sh;
    const uint32_t total = suf[0];
    __syncthreads();
    if (digit <= 0 && total <= (uint32_t)S.k_rem) {              // fewer valid elements than k: take all
        if (t == 0) { S_sh->done = 1; S_sh->n_sel = (int32_t)total; }
        __syncthreads();
        return;
    }
    const uint32_t need = (uint32_t)S.k_rem;
    if (suf[t] >= need && suf[t + 1] < need) {                   // exactly one thread
        uint32_t above = suf[t + 1];
        int b = 4 * t + 3;
        for (; b > 4 * t; --b) {
            if (above + h[b] >= need) break;
            above += h[b];
        }
        SelState N = S;
        const uint32_t superset = (uint32_t)S.n_above + above + h[b];
        N.n_sel = k;
        if (digit < 0) {
            if (b == 0 || b == MSR_SEL_BINS - 1) {               // a clamped bin: no prefix is known -- the general path, from scratch
                N.done = 0;
            } else {
                N.pref_hi = (win_base + (uint64_t)b) << WIN_SHIFT;
                N.mask_hi = ~(uint64_t)0 << WIN_SHIFT;
                N.n_above += (int32_t)above;
                N.k_rem -= (int32_t)above;
                if (superset <= MSR_SEL_CAP) N.done = 1;
            }
        } else {
            int part, shift, width;
            digit_pos<SB>(digit, part, shift, width);
            const uint64_t wmask = ((uint64_t)1 << width) - 1;
            if (part == 0) { N.pref_hi |= (uint64_t)b << shift; N.mask_hi |= wmask << shift; }
            else { N.pref_lo |= (uint32_t)b << shift; N.mask_lo |= (uint32_t)(wmask << shift); }
            N.n_above += (int32_t)above;
            N.k_rem -= (int32_t)above;
            if (superset <= MSR_SEL_CAP || digit == KeyCfg<SB>::ND - 1) N.done = 1;
        }
        if (superset_out) *superset_out = (int)superset;
        *S_sh = N;
    }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ bool key_of(const T s, int64_t i, uint64_t& khi, uint32_t& klo) {
    if (!msr_valid(s)) return false;
    khi = ScoreTraits<T>::ord(s);
    klo = ~(uint32_t)i;
    return true;
}

// A row is either dense (element i belongs to index i, n elements) or a LIST (element i belongs to index idx[i]) cut into
// n_seg SEGMENTS: segment s of row q holds counts[q * n_seg + s] elements, in no particular order, from position
// s * seg_stride of the row.  Lists are what the BM25 kernel emits -- only the documents that are candidates at all, one
// segment per span of document tiles, written without any atomic (every segment has exactly one writer).
struct RowView {
    const int32_t* idx;      // null: dense
    const int32_t* counts;   // null: every row has n elements
    int32_t n_seg;           // lists: segments per row (dense: unused)
    int64_t seg_stride;      // lists: elements between the starts of consecutive segments
    const uint64_t* win_base;  // non-null (64-bit scores only): pass -1 bins the 20-bit key prefixes, see msr_internal.h
};

__device__ __forceinline__ int64_t row_index(const int32_t* idx_row, int64_t i) { return idx_row ? idx_row[i] : i; }
// Work split: part `part` of `parts` takes a contiguous range of a dense row, or whole segments of a list.
__device__ __forceinline__ void part_segments(const RowView& v, int part, int parts, int& s_first, int& s_last) {
    if (!v.counts) { s_first = 0; s_last = 1; return; }
    const int per = (v.n_seg + parts - 1) / parts;
    s_first = part * per;
    s_last = s_first + per < v.n_seg ? s_first + per : v.n_seg;
}
__device__ __forceinline__ void segment_range(const RowView& v, int q, int64_t n_dense, int s, int part, int parts,
                                              int64_t& lo, int64_t& hi) {
    if (!v.counts) {
        const int64_t per = (n_dense + parts - 1) / parts;
        lo = (int64_t)part * per;
        hi = lo + per < n_dense ? lo + per : n_dense;
    } else {
        lo = (int64_t)s * v.seg_stride;
        hi = lo + v.counts[(int64_t)q * v.n_seg + s];
    }
}

// Histogram of digit `digit` over the elements that match the resolved prefix.  digit 0 needs no state.
template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void sel_hist_kernel(const T* __restrict__ scores, int64_t n_dense,
                                                                int64_t stride, RowView view, int digit,
                                                                const SelState* __restrict__ st,
                                                                uint32_t* __restrict__ hist, const int32_t* __restrict__ gate,
                                                                int gate_per64) {
    constexpr int SB = ScoreTraits<T>::SB;
    const int q = blockIdx.y;
    if (gate && gate[gate_per64 ? q >> 6 : 0] == 0) return;      // a fallback launch that is not needed (for this query's slice)
    SelState S;
    S.pref_hi = S.mask_hi = 0; S.pref_lo = S.mask_lo = 0; S.done = 0;
    if (digit > 0) {
        S = st[q];
        if (S.done) return;
    }
    __shared__ uint32_t h[MSR_SEL_BINS];
    for (int b = threadIdx.x; b < MSR_SEL_BINS; b += SEL_THREADS) h[b] = 0;
    __syncthreads();
    const bool window = digit < 0;
    const int64_t wbase = window ? (int64_t)view.win_base[q] : 0;
    int part, shift, width;
    digit_pos<SB>(window ? 0 : digit, part, shift, width);
    const uint32_t wmask = (1u << width) - 1u;
    const T* row = scores + (int64_t)q * stride;
    // (a pass over the score part of the key with no index bits resolved does not look at the index: a list's 4 bytes of
    // index per element are then not loaded -- a third of the pass' bytes)
    const bool need_index = part != 0 || S.mask_lo != 0;
    const int32_t* irow = view.idx && need_index ? view.idx + (int64_t)q * stride : nullptr;
    int s_first, s_last;
    part_segments(view, (int)blockIdx.x, (int)gridDim.x, s_first, s_last);
    for (int sg = s_first; sg < s_last; ++sg) {
        int64_t lo, hi;
        segment_range(view, q, n_dense, sg, (int)blockIdx.x, (int)gridDim.x, lo, hi);
        // four independent loads in flight per thread (one per iteration left the pass latency-bound)
        for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * SEL_THREADS) {
            T v[4];
            int64_t ix[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + (int64_t)u * SEL_THREADS;
                v[u] = i < hi ? row[i] : ScoreTraits<T>::neg_inf();
                ix[u] = i < hi ? row_index(irow, i) : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint64_t khi; uint32_t klo;
                if (!key_of(v[u], ix[u], khi, klo)) continue;
                if ((khi & S.mask_hi) != S.pref_hi || (klo & S.mask_lo) != S.pref_lo) continue;
                uint32_t dg = part == 0 ? (uint32_t)(khi >> shift) & wmask : (klo >> shift) & wmask;
                if (window) {
                    const int64_t b = (int64_t)(khi >> WIN_SHIFT) - wbase;
                    dg = b < 0 ? 0u : (b > MSR_SEL_BINS - 1 ? (uint32_t)(MSR_SEL_BINS - 1) : (uint32_t)b);
                }
                atomicAdd(&h[dg], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* gh = hist + (int64_t)q * MSR_SEL_BINS;
    for (int b = threadIdx.x; b < MSR_SEL_BINS; b += SEL_THREADS)
        if (h[b]) atomicAdd(&gh[b], h[b]);
}

// One workgroup per query: consume the global histogram of `digit` (and leave it zeroed), update the state.
template <int SB>
__global__ __launch_bounds__(SCAN_THREADS) void sel_scan_kernel(SelState* __restrict__ st,
                                                                 uint32_t* __restrict__ hist, int digit, int k,
                                                                 const int32_t* __restrict__ gate, int gate_per64,
                                                                 const uint64_t* __restrict__ win_base) {
    if (gate && gate[gate_per64 ? (int)blockIdx.x >> 6 : 0] == 0) return;
    __shared__ uint32_t h[MSR_SEL_BINS];
    __shared__ uint32_t suf[SCAN_THREADS + 1];
    __shared__ SelState S_sh;
    const int q = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        if (digit <= 0) {
            SelState S;
            S.pref_hi = S.mask_hi = 0; S.pref_lo = S.mask_lo = 0;
            S.k_rem = k; S.n_above = 0; S.done = 0; S.n_sel = 0;
            S_sh = S;
        } else {
            S_sh = st[q];
        }
    }
    __syncthreads();
    if (S_sh.done) return;                                       // nothing was added to hist[q] in this pass
    uint32_t* gh = hist + (int64_t)q * MSR_SEL_BINS;
    for (int j = 0; j < 4; ++j) {
        h[4 * t + j] = gh[4 * t + j];
        gh[4 * t + j] = 0;
    }
    __syncthreads();
    select_step<SB>(h, suf, &S_sh, digit, k, nullptr, digit < 0 ? win_base[q] : 0);
    if (t == 0) st[q] = S_sh;
}

// Compaction of the elements at or above the resolved prefix.  Matches are staged in LDS and appended to the
// query's candidate list with ONE global atomic per workgroup (same-address atomics from every lane made
// this the slowest kernel of the select in the first profile).  Queries that two passes could not resolve
// (superset still > MSR_SEL_CAP: huge tie groups) are left to the final kernel's in-kernel loop.
template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void sel_compact_kernel(const T* __restrict__ scores, int64_t n_dense,
                                                                   int64_t stride, RowView view,
                                                                   const SelState* __restrict__ st,
                                                                   uint64_t* __restrict__ cand_hi,
                                                                   uint32_t* __restrict__ cand_lo,
                                                                   int32_t* __restrict__ cand_n,
                                                                   const int32_t* __restrict__ gate, int gate_per64) {
    if (gate && gate[gate_per64 ? (int)blockIdx.y >> 6 : 0] == 0) return;
    constexpr int STAGE = 1024;                                  // staged matches per workgroup (12 KB of LDS)
    __shared__ uint64_t s_hi[STAGE];
    __shared__ uint32_t s_lo[STAGE];
    __shared__ int s_n, s_base;
    const int q = blockIdx.y;
    const SelState S = st[q];
    if (!S.done) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const T* row = scores + (int64_t)q * stride;
    const int32_t* irow = view.idx ? view.idx + (int64_t)q * stride : nullptr;
    int s_first, s_last;
    part_segments(view, (int)blockIdx.x, (int)gridDim.x, s_first, s_last);
    for (int sg = s_first; sg < s_last; ++sg) {
        int64_t lo, hi;
        segment_range(view, q, n_dense, sg, (int)blockIdx.x, (int)gridDim.x, lo, hi);
        for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * SEL_THREADS) {
            T v[4];
            int64_t ix[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + (int64_t)u * SEL_THREADS;
                v[u] = i < hi ? row[i] : ScoreTraits<T>::neg_inf();
                ix[u] = i < hi ? row_index(irow, i) : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint64_t khi; uint32_t klo;
                if (!key_of(v[u], ix[u], khi, klo)) continue;
                const uint64_t mh = khi & S.mask_hi;
                const bool ge = mh > S.pref_hi || (mh == S.pref_hi && (klo & S.mask_lo) >= S.pref_lo);
                if (!ge) continue;
                const int pos = atomicAdd(&s_n, 1);              // LDS atomic
                if (pos < STAGE) {
                    s_hi[pos] = khi; s_lo[pos] = klo;
                } else {                                         // more matches than the stage holds: append directly
                    const int g = atomicAdd(&cand_n[q], 1);
                    if (g < MSR_SEL_CAP) {
                        cand_hi[(int64_t)q * MSR_SEL_CAP + g] = khi;
                        cand_lo[(int64_t)q * MSR_SEL_CAP + g] = klo;
                    }
                }
            }
        }
    }
    __syncthreads();
    int cnt = s_n;
    if (cnt > STAGE) cnt = STAGE;
    if (cnt == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&cand_n[q], cnt);
    __syncthreads();
    const int base = s_base;
    for (int j = threadIdx.x; j < cnt; j += SEL_THREADS) {
        if (base + j < MSR_SEL_CAP) {
            cand_hi[(int64_t)q * MSR_SEL_CAP + base + j] = s_hi[j];
            cand_lo[(int64_t)q * MSR_SEL_CAP + base + j] = s_lo[j];
        }
    }
}

// Bitonic sort of (hi, lo) keys, descending, in LDS.  P is a power of two.
__device__ __forceinline__ bool key_less(uint64_t ah, uint32_t al, uint64_t bh, uint32_t bl) {
    return ah < bh || (ah == bh && al < bl);
}

using msr_sort::stage_sync;

// One workgroup per query: exact sort of the candidates and output.  If the two streaming passes did not
// resolve the query (state not done), this workgroup finishes the radix select on its own over the score
// row -- slow (one CU reads the row once per remaining digit) but exact, and only reached with tie groups
// larger than MSR_SEL_CAP.
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void sel_final_kernel(const T* __restrict__ scores, int64_t n_dense,
                                                                  int64_t stride, RowView view,
                                                                  SelState* __restrict__ st,
                                                                  const uint64_t* __restrict__ cand_hi,
                                                                  const uint32_t* __restrict__ cand_lo,
                                                                  int32_t* __restrict__ cand_n, int k,
                                                                  int32_t* __restrict__ out_doc,
                                                                  T* __restrict__ out_score,
                                                                  int32_t* __restrict__ out_n,
                                                                  const int32_t* __restrict__ gate, int gate_per64) {
    constexpr int SB = ScoreTraits<T>::SB;
    if (gate && gate[gate_per64 ? (int)blockIdx.x >> 6 : 0] == 0) return;
    __shared__ uint64_t khi[MSR_SEL_CAP];
    __shared__ uint32_t klo[MSR_SEL_CAP];
    __shared__ uint32_t suf[SCAN_THREADS + 1];
    __shared__ SelState S_sh;
    __shared__ int s_cnt;
    const int q = blockIdx.x, t = threadIdx.x;
    const T* row = scores + (int64_t)q * stride;
    const int32_t* irow = view.idx ? view.idx + (int64_t)q * stride : nullptr;
    int n_sg = 1, sg0 = 0;
    part_segments(view, 0, 1, sg0, n_sg);                        // (this workgroup walks every segment of the row)
    DBG_TS(0);
    if (t == 0) S_sh = st[q];
    __syncthreads();
    int cnt;
    DBG_TS(1);
    if (S_sh.done) {
        cnt = cand_n[q];
        if (cnt > MSR_SEL_CAP) cnt = MSR_SEL_CAP;
        const uint64_t* ch = cand_hi + (int64_t)q * MSR_SEL_CAP;
        const uint32_t* cl = cand_lo + (int64_t)q * MSR_SEL_CAP;
        int Pk = 64;
        while (Pk < k) Pk <<= 1;
        if (cnt <= Pk) {
            for (int i = t; i < cnt; i += SCAN_THREADS) { khi[i] = ch[i]; klo[i] = cl[i]; }
        } else {
            // More candidates than the sort needs slots for k (the bin of the k-th key at 24 resolved bits holds a few dozen
            // elements: 1000 + 30 candidates would be sorted as 2048): resolve further digits HERE, on the candidates (L2-resident,
            // <= 48 KB), until they fit the next power of two above k -- the sort is 2.4 x shorter at 1024 than at 2048 entries.
            uint32_t* h = (uint32_t*)khi;                        // (the key array is not loaded yet)
            int d = 0;
            for (; d < KeyCfg<SB>::ND; ++d) {                    // first digit that is not resolved
                int part, shift, width;
                digit_pos<SB>(d, part, shift, width);
                if (!(part == 0 ? (S_sh.mask_hi >> shift) & 1 : (S_sh.mask_lo >> shift) & 1)) break;
            }
            int cur = cnt;
            for (; cur > Pk && d < KeyCfg<SB>::ND; ++d) {
                for (int b = t; b < MSR_SEL_BINS; b += SCAN_THREADS) h[b] = 0;
                __syncthreads();
                const SelState S = S_sh;
                int part, shift, width;
                digit_pos<SB>(d, part, shift, width);
                const uint32_t wmask = (1u << width) - 1u;
                for (int i = t; i < cnt; i += SCAN_THREADS) {
                    const uint64_t a = ch[i];
                    const uint32_t b = cl[i];
                    if ((a & S.mask_hi) != S.pref_hi || (b & S.mask_lo) != S.pref_lo) continue;
                    atomicAdd(&h[part == 0 ? (uint32_t)(a >> shift) & wmask : (b >> shift) & wmask], 1u);
                }
                __syncthreads();
                select_step<SB>(h, suf, &S_sh, d, k, &s_cnt);
                cur = s_cnt;
                __syncthreads();
            }
            DBG_TS(2);
            if (t == 0) s_cnt = 0;
            __syncthreads();
            const SelState S = S_sh;
            for (int i = t; i < cnt; i += SCAN_THREADS) {
                const uint64_t a = ch[i];
                const uint32_t b = cl[i];
                const uint64_t mh = a & S.mask_hi;
                if (mh > S.pref_hi || (mh == S.pref_hi && (b & S.mask_lo) >= S.pref_lo)) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    khi[pos] = a; klo[pos] = b;                  // (a subset of the cnt <= MSR_SEL_CAP candidates)
                }
            }
            __syncthreads();
            cnt = s_cnt;
        }
    } else {
        uint32_t* h = (uint32_t*)khi;                            // the histogram lives in the (still unused) key array
        int d0 = 0;
        for (; d0 < KeyCfg<SB>::ND; ++d0) {                      // first digit that is not (fully) resolved: 2 after the two
            int part, shift, width;                              // streaming passes, 1 after a window pass, 0 if that gave up
            digit_pos<SB>(d0, part, shift, width);
            if (!(part == 0 ? (S_sh.mask_hi >> shift) & 1 : (S_sh.mask_lo >> shift) & 1)) break;
        }
        for (int d = d0; d < KeyCfg<SB>::ND; ++d) {
            for (int b = t; b < MSR_SEL_BINS; b += SCAN_THREADS) h[b] = 0;
            __syncthreads();
            const SelState S = S_sh;
            int part, shift, width;
            digit_pos<SB>(d, part, shift, width);
            const uint32_t wmask = (1u << width) - 1u;
            for (int sg = 0; sg < n_sg; ++sg) {
                int64_t lo, hi;
                segment_range(view, q, n_dense, sg, 0, 1, lo, hi);
                for (int64_t i = lo + t; i < hi; i += SCAN_THREADS) {
                    uint64_t a; uint32_t b;
                    if (!key_of(row[i], row_index(irow, i), a, b)) continue;
                    if ((a & S.mask_hi) != S.pref_hi || (b & S.mask_lo) != S.pref_lo) continue;
                    atomicAdd(&h[part == 0 ? (uint32_t)(a >> shift) & wmask : (b >> shift) & wmask], 1u);
                }
            }
            __syncthreads();
            select_step<SB>(h, suf, &S_sh, d, k);
            if (S_sh.done) break;
        }
        if (t == 0) s_cnt = 0;
        __syncthreads();
        const SelState S = S_sh;
        for (int sg = 0; sg < n_sg; ++sg) {
            int64_t lo, hi;
            segment_range(view, q, n_dense, sg, 0, 1, lo, hi);
            for (int64_t i = lo + t; i < hi; i += SCAN_THREADS) {
                uint64_t a; uint32_t b;
                if (!key_of(row[i], row_index(irow, i), a, b)) continue;
                const uint64_t mh = a & S.mask_hi;
                if (mh > S.pref_hi || (mh == S.pref_hi && (b & S.mask_lo) >= S.pref_lo)) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < MSR_SEL_CAP) { khi[pos] = a; klo[pos] = b; }   // h is dead: safe to overwrite
                }
            }
        }
        __syncthreads();
        cnt = s_cnt < MSR_SEL_CAP ? s_cnt : MSR_SEL_CAP;
    }
    int P = 64;
    while (P < cnt) P <<= 1;
    __syncthreads();
    DBG_TS(3);
    for (int i = cnt + t; i < P; i += SCAN_THREADS) { khi[i] = 0; klo[i] = 0; }
    __syncthreads();
    msr_sort::bitonic_sort<SCAN_THREADS, false>(khi, klo, nullptr, P, false);
    DBG_TS(4);
    int n_sel = S_sh.n_sel;
    if (n_sel > cnt) n_sel = cnt;
    for (int i = t; i < k; i += SCAN_THREADS) {
        const bool ok = i < n_sel;
        out_doc[(int64_t)q * k + i] = ok ? (int32_t)~klo[i] : -1;
        out_score[(int64_t)q * k + i] = ok ? ScoreTraits<T>::unord(khi[i]) : ScoreTraits<T>::neg_inf();
    }
    if (t == 0) {
        out_n[q] = n_sel;
        cand_n[q] = 0;                                           // invariant: zero between calls
    }
    DBG_TS(5);
#ifdef MSR_DIAG
    if (blockIdx.x == 0 && t == 0) { msr_dbg_ts[6] = (unsigned long long)cnt; msr_dbg_ts[7] = (unsigned long long)P; }
#endif
}

template <typename T>
hipError_t select_impl(const T* scores, int64_t n, int64_t stride, RowView view, int nq, int k, const SelScratch& sc,
                       int32_t* out_doc, T* out_score, int32_t* out_n, hipStream_t stream) {
    constexpr int SB = ScoreTraits<T>::SB;
    if (nq <= 0) return hipSuccess;
    int64_t parts = view.counts ? view.n_seg : (n + 8191) / 8192;     // lists: whole segments per workgroup
    const int64_t max_parts = 2048 / nq > 0 ? 2048 / nq : 1;
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    dim3 grid((unsigned)parts, (unsigned)nq);
    // two streaming histogram passes (24 key bits) -- or ONE over a window of 20-bit prefixes --, one compaction, one exact
    // sort: 6 (4) launches
    const bool window = SB == 64 && view.win_base != nullptr;
    for (int d = window ? -1 : 0; d < (window ? 0 : 2); ++d) {
        sel_hist_kernel<T><<<grid, SEL_THREADS, 0, stream>>>(scores, n, stride, view, d, sc.state, sc.hist, sc.gate, sc.gate_per64);
        sel_scan_kernel<SB><<<nq, SCAN_THREADS, 0, stream>>>(sc.state, sc.hist, d, k, sc.gate, sc.gate_per64, view.win_base);
    }
    sel_compact_kernel<T><<<grid, SEL_THREADS, 0, stream>>>(scores, n, stride, view, sc.state, sc.cand_hi, sc.cand_lo,
                                                             sc.cand_n, sc.gate, sc.gate_per64);
    sel_final_kernel<T><<<nq, SCAN_THREADS, 0, stream>>>(scores, n, stride, view, sc.state, sc.cand_hi, sc.cand_lo,
                                                          sc.cand_n, k, out_doc, out_score, out_n, sc.gate, sc.gate_per64);
    return hipGetLastError();
}

// ---- merge of per-shard lists ---------------------------------------------------------------------
// Merge of n_parts lists of <= k records, EACH SORTED by (score descending, index ascending) -- what every *_topk entry
// point returns -- into the k best.  One workgroup per query.  The lists sit side by side in LDS (Pk = k rounded up to a power
// of two entries each, missing lists and tails filled with the smallest key) and are reduced pairwise: the element-wise
// maximum of list A and the REVERSED list B is a bitonic sequence that holds the Pk best of both, log2 Pk compare-exchange
// stages put it in order again.  log2(parts) rounds on half as many lists each: 33 stages over 5 k + 2.5 k + 1.3 k entries
// for 8 x 1000 records instead of the 91 stages over 8192 entries of a full sort.
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void merge_kernel(const int32_t* __restrict__ in_doc,
                                                              const T* __restrict__ in_score,
                                                              const int32_t* __restrict__ in_n,
                                                              const int32_t* __restrict__ in_pay, int n_parts,
                                                              int64_t pstride, int nq, int k, int32_t* __restrict__ out_doc,
                                                              T* __restrict__ out_score,
                                                              int32_t* __restrict__ out_n,
                                                              int32_t* __restrict__ out_pay, int only_left_over) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q = blockIdx.x;
    if (only_left_over && out_n[q] != -1) return;                 // merge_rank_kernel has written this query's result
    int lg = 6;
    while ((1 << lg) < k) ++lg;
    const int Pk = 1 << lg;                                       // entries per list
    int NP = 1;
    while (NP < n_parts) NP <<= 1;                                // lists, phantom ones included
    const int P = NP * Pk;
    uint64_t* khi = (uint64_t*)smem;
    uint32_t* klo = (uint32_t*)(smem + (size_t)P * 8);
    int32_t* pay = (int32_t*)(smem + (size_t)P * 12);            // (only touched when in_pay is given)
    __shared__ int n_valid;
    // part p of every array starts pstride BYTES after part p - 1 (pstride = 0: the parts are contiguous arrays)
    auto part = [&](const auto* base, int p, int64_t elems) {
        typedef decltype(base) PT;
        return pstride ? (PT)((const char*)base + (int64_t)p * pstride) : base + (int64_t)p * elems;
    };
    if (threadIdx.x == 0) {
        int s = 0;
        for (int p = 0; p < n_parts; ++p) {
            int c = part(in_n, p, nq)[q];
            s += c < 0 ? 0 : (c > k ? k : c);
        }
        n_valid = s;
    }
    for (int i = threadIdx.x; i < P; i += SCAN_THREADS) {
        uint64_t h = 0; uint32_t l = 0; int32_t v = -1;
        const int p = i >> lg, r = i & (Pk - 1);
        if (p < n_parts && r < k) {
            int c = part(in_n, p, nq)[q];
            if (r < c) {
                const int64_t off = (int64_t)q * k + r;
                const T s = part(in_score, p, (int64_t)nq * k)[off];
                if (msr_valid(s)) {
                    h = ScoreTraits<T>::ord(s);
                    l = ~(uint32_t)part(in_doc, p, (int64_t)nq * k)[off];
                    if (in_pay) v = part(in_pay, p, (int64_t)nq * k)[off];
                }
            }
        }
        khi[i] = h; klo[i] = l;
        if (in_pay) pay[i] = v;
    }
    __syncthreads();
    for (int step = 1; step < NP; step <<= 1) {                   // lists a = 2 m step and b = a + step -> a
        const int n_pairs = NP / (2 * step);
        for (int idx = threadIdx.x; idx < (n_pairs << lg); idx += SCAN_THREADS) {
            const int m = idx >> lg, i = idx & (Pk - 1);
            const int ai = ((2 * m * step) << lg) + i, bi = ((2 * m * step + step) << lg) + (Pk - 1 - i);
            const uint64_t ah = khi[ai], bh = khi[bi];
            const uint32_t al = klo[ai], bl = klo[bi];
            if (key_less(ah, al, bh, bl)) {
                khi[ai] = bh; klo[ai] = bl;
                if (in_pay) pay[ai] = pay[bi];
            }
        }
        __syncthreads();
        for (int j = Pk >> 1; j > 0; j >>= 1) {
            for (int idx = threadIdx.x; idx < (n_pairs << (lg - 1)); idx += SCAN_THREADS) {
                const int m = idx >> (lg - 1), t = idx & ((Pk >> 1) - 1);
                const int a0 = ((2 * m * step) << lg) + (((t & ~(j - 1)) << 1) | (t & (j - 1))), b0 = a0 + j;
                const uint64_t ah = khi[a0], bh = khi[b0];
                const uint32_t al = klo[a0], bl = klo[b0];
                if (key_less(ah, al, bh, bl)) {
                    khi[a0] = bh; khi[b0] = ah; klo[a0] = bl; klo[b0] = al;
                    if (in_pay) { const int32_t v = pay[a0]; pay[a0] = pay[b0]; pay[b0] = v; }
                }
            }
            stage_sync(j, j > 1 ? j >> 1 : Pk);                   // (after j = 1: the next round's maximum stage, or the output)
        }
    }
    const int n_sel = n_valid < k ? n_valid : k;
    for (int i = threadIdx.x; i < k; i += SCAN_THREADS) {
        const bool ok = i < n_sel && !(khi[i] == 0 && klo[i] == 0);
        out_doc[(int64_t)q * k + i] = ok ? (int32_t)~klo[i] : -1;
        out_score[(int64_t)q * k + i] = ok ? ScoreTraits<T>::unord(khi[i]) : ScoreTraits<T>::neg_inf();
        if (out_pay) out_pay[(int64_t)q * k + i] = ok && in_pay ? pay[i] : -1;
    }
    if (threadIdx.x == 0) out_n[q] = n_sel;
}

// The merge most calls take.  Of N sorted lists of <= k records only a prefix of each can reach the k best: with `quota` =
// ceil(k / lists that hold at least that many), the smallest of those lists' quota-th keys -- the CUT -- has at least k keys at
// or above it, so nothing below it is wanted (8 lists of 1000 from statistically alike shards: ~130-200 records of each, 1.2 k
// of the 8 k).  The prefixes (<= MRG_CAP / lists records each) are staged in LDS and every record finds its place in the output
// by counting: its index in its own list + for every other list the number of keys above it (binary search; on equal keys --
// the same document in two lists -- the lower list number goes first).  No compare-exchange network, no barrier after the
// staging.  A query whose prefixes do not fit (lists from very unlike shards), or with an invalid score in them, is left to
// the merge tree below: out_n[q] = -1 is the message.
constexpr int MRG_THREADS = 512;
constexpr int MRG_CAP = 2048;
constexpr int MRG_MAX_PARTS = 64;

template <typename T>
__global__ __launch_bounds__(MRG_THREADS) void merge_rank_kernel(const int32_t* __restrict__ in_doc,
                                                                  const T* __restrict__ in_score,
                                                                  const int32_t* __restrict__ in_n,
                                                                  const int32_t* __restrict__ in_pay, int n_parts,
                                                                  int64_t pstride, int nq, int k, int32_t* __restrict__ out_doc,
                                                                  T* __restrict__ out_score, int32_t* __restrict__ out_n,
                                                                  int32_t* __restrict__ out_pay) {
    __shared__ uint64_t khi[MRG_CAP];
    __shared__ uint32_t klo[MRG_CAP];
    __shared__ int32_t pay[MRG_CAP];
    __shared__ int s_c[MRG_MAX_PARTS], s_len[MRG_MAX_PARTS];
    __shared__ int s_quota, s_flag;
    const int q = blockIdx.x, t = threadIdx.x;
    int lgS = 11, NP = 1;
    while (NP < n_parts) { NP <<= 1; --lgS; }
    const int SPEC = 1 << lgS;                                    // staged records per list
    auto part = [&](const auto* base, int p, int64_t elems) {
        typedef decltype(base) PT;
        return pstride ? (PT)((const char*)base + (int64_t)p * pstride) : base + (int64_t)p * elems;
    };
    // ONE round trip to memory: every thread asks for its list's count and its records at once (a list is k records long
    // whatever its count: what lies behind the count is masked afterwards)
    if (t == 0) s_flag = 0;
    if (t < MRG_MAX_PARTS) s_c[t] = 0;
    __syncthreads();
    constexpr int PER = MRG_CAP / MRG_THREADS;                     // staged records per thread: all requested before any is used
    int c_[PER], d_[PER], pv_[PER];
    T sc_[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = t + j * MRG_THREADS, p = i >> lgS, r = i & (SPEC - 1);
        const bool in = p < n_parts && r < k;
        const int64_t off = (int64_t)q * k + (in ? r : 0);
        const int pp = in ? p : 0;
        c_[j] = in ? part(in_n, pp, nq)[q] : 0;
        sc_[j] = part(in_score, pp, (int64_t)nq * k)[off];
        d_[j] = part(in_doc, pp, (int64_t)nq * k)[off];
        pv_[j] = in_pay ? part(in_pay, pp, (int64_t)nq * k)[off] : -1;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = t + j * MRG_THREADS, p = i >> lgS, r = i & (SPEC - 1);
        uint64_t h = 0; uint32_t l = 0; int32_t v = -1;
        const int c = c_[j] < 0 ? 0 : (c_[j] > k ? k : c_[j]);
        if (p < n_parts && r == 0) s_c[p] = c;
        if (r < c) {                                              // (c = 0 for the slots outside the lists)
            if (msr_valid(sc_[j])) { h = ScoreTraits<T>::ord(sc_[j]); l = ~(uint32_t)d_[j]; v = pv_[j]; }
            else s_flag = 1;
        }
        khi[i] = h; klo[i] = l; pay[i] = v;
    }
    __syncthreads();
    if (t == 0) {                                                 // quota: ceil(k / #lists with at least quota records), fixed point
        int n_long = n_parts, quota = 0;
        for (int it = 0; it <= n_parts && n_long > 0; ++it) {
            quota = (k + n_long - 1) / n_long;
            int m = 0;
            for (int p = 0; p < n_parts; ++p) m += s_c[p] >= quota;
            if (m == n_long) break;
            n_long = m;
        }
        s_quota = n_long > 0 ? quota : 0;                         // 0: no cut (fewer than k records in all, or nearly)
        if (s_quota > SPEC) s_flag = 1;                           // (the long lists' wanted prefixes are longer than what is staged)
    }
    __syncthreads();
    const int quota = s_quota;
    // records of list p with key > (or, with_equal, >=) the given key, among its first n
    auto count_above = [&](int p, int n, uint64_t kh, uint32_t kl, bool with_equal) {
        int lo = 0, hi = n;
        const int base = p << lgS;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const uint64_t mh = khi[base + mid];
            const uint32_t ml = klo[base + mid];
            const bool above = mh > kh || (mh == kh && (with_equal ? ml >= kl : ml > kl));
            if (above) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    if (t < MRG_MAX_PARTS && !s_flag) {
        int len = 0;
        if (t < n_parts) {
            const int c = s_c[t], staged = c < SPEC ? c : SPEC;
            len = staged;
            if (quota > 0) {
                uint64_t ch = ~(uint64_t)0; uint32_t cl = ~0u;    // the cut: smallest quota-th key of the long lists
                for (int p = 0; p < n_parts; ++p) {
                    const int at = (p << lgS) + quota - 1;
                    if (s_c[p] >= quota && key_less(khi[at], klo[at], ch, cl)) { ch = khi[at]; cl = klo[at]; }
                }
                len = count_above(t, staged, ch, cl, true);
            }
            if (len == SPEC && c > SPEC) s_flag = 1;              // the wanted prefix may be longer than what is staged
        }
        s_len[t] = len;
    }
    __syncthreads();
    if (s_flag) {
        if (t == 0) out_n[q] = -1;
        return;
    }
    int total = 0;
    for (int p = 0; p < n_parts; ++p) total += s_len[p];
    const int n_sel = total < k ? total : k;
    // record e of the `total` kept ones = record r of list p; its rank: r + what the other lists hold above it.  The searches
    // in four other lists run side by side (independent LDS round trips).
    for (int e = t; e < total; e += MRG_THREADS) {
        int p = 0, r = e;
        while (r >= s_len[p]) { r -= s_len[p]; ++p; }
        const int i = (p << lgS) + r;
        const uint64_t kh = khi[i];
        const uint32_t kl = klo[i];
        int rank = r;
        for (int o0 = 0; o0 < n_parts; o0 += 4) {
            int lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = o0 + u;
                lo[u] = 0;
                hi[u] = o < n_parts && o != p ? s_len[o] : 0;
            }
            for (int st = 0; st <= lgS; ++st) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (lo[u] < hi[u]) {
                        const int mid = (lo[u] + hi[u]) >> 1, at = ((o0 + u) << lgS) + mid;
                        const uint64_t mh = khi[at];
                        const uint32_t ml = klo[at];
                        const bool above = mh > kh || (mh == kh && (o0 + u < p ? ml >= kl : ml > kl));
                        if (above) lo[u] = mid + 1; else hi[u] = mid;
                    }
                }
            }
            rank += lo[0] + lo[1] + lo[2] + lo[3];
        }
        if (rank < k) {
            const bool ok = !(kh == 0 && kl == 0);
            out_doc[(int64_t)q * k + rank] = ok ? (int32_t)~kl : -1;
            out_score[(int64_t)q * k + rank] = ok ? ScoreTraits<T>::unord(kh) : ScoreTraits<T>::neg_inf();
            if (out_pay) out_pay[(int64_t)q * k + rank] = ok && in_pay ? pay[i] : -1;
        }
    }
    for (int i = n_sel + t; i < k; i += MRG_THREADS) {
        out_doc[(int64_t)q * k + i] = -1;
        out_score[(int64_t)q * k + i] = ScoreTraits<T>::neg_inf();
        if (out_pay) out_pay[(int64_t)q * k + i] = -1;
    }
    if (t == 0) out_n[q] = n_sel;
}

}  // namespace

hipError_t msr_select_topk(int score_bits, const void* scores, int64_t n, int64_t stride, int nq, int k,
                           const SelScratch& sc, int32_t* out_doc, void* out_score, int32_t* out_n,
                           hipStream_t stream) {
    const RowView dense{nullptr, nullptr, 1, 0, nullptr};
    if (score_bits == 32)
        return select_impl<float>((const float*)scores, n, stride, dense, nq, k, sc, out_doc, (float*)out_score, out_n, stream);
    return select_impl<double>((const double*)scores, n, stride, dense, nq, k, sc, out_doc, (double*)out_score, out_n, stream);
}

hipError_t msr_select_topk_list(const double* scores, const int32_t* idx, const int32_t* counts, int n_seg, int64_t seg_stride,
                                int64_t stride, int nq, int k, const SelScratch& sc, int32_t* out_doc,
                                double* out_score, int32_t* out_n, hipStream_t stream, const uint64_t* win_base) {
    if (n_seg < 1 || !counts || !idx) return hipErrorInvalidValue;
    const RowView list{idx, counts, n_seg, seg_stride, win_base};
    return select_impl<double>(scores, stride, stride, list, nq, k, sc, out_doc, out_score, out_n, stream);
}

hipError_t msr_merge_lists(int score_bits, const int32_t* in_doc, const void* in_score, const int32_t* in_n,
                           const int32_t* in_pay, int n_parts, int64_t part_stride_bytes, int nq, int k, int32_t* out_doc,
                           void* out_score, int32_t* out_n, int32_t* out_pay, hipStream_t stream) {
    if (n_parts < 1 || k < 1) return hipErrorInvalidValue;
    int Pk = 64, NP = 1;                                // the kernel's layout: NP lists of Pk entries
    while (Pk < k) Pk <<= 1;
    while (NP < n_parts) NP <<= 1;
    const size_t lds = (size_t)NP * Pk * (in_pay ? 16 : 12);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    // the counting merge for every query it can take (out_n[q] = -1 where it cannot), the merge tree for the rest
    const int fast = n_parts <= MRG_MAX_PARTS && n_parts > 1;
    if (score_bits == 32) {
        if (fast)
            merge_rank_kernel<float><<<nq, MRG_THREADS, 0, stream>>>(in_doc, (const float*)in_score, in_n, in_pay, n_parts,
                                                                     part_stride_bytes, nq, k, out_doc, (float*)out_score, out_n, out_pay);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)merge_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        merge_kernel<float><<<nq, SCAN_THREADS, lds, stream>>>(in_doc, (const float*)in_score, in_n, in_pay, n_parts,
                                                               part_stride_bytes, nq, k, out_doc, (float*)out_score, out_n, out_pay, fast);
    } else {
        if (fast)
            merge_rank_kernel<double><<<nq, MRG_THREADS, 0, stream>>>(in_doc, (const double*)in_score, in_n, in_pay, n_parts,
                                                                      part_stride_bytes, nq, k, out_doc, (double*)out_score, out_n, out_pay);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)merge_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        merge_kernel<double><<<nq, SCAN_THREADS, lds, stream>>>(in_doc, (const double*)in_score, in_n, in_pay, n_parts,
                                                                part_stride_bytes, nq, k, out_doc, (double*)out_score, out_n, out_pay, fast);
    }
    return hipGetLastError();
}
