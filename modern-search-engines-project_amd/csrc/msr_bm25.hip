// K1 -- BM25 term-at-a-time scoring over HBM-resident CSR postings (gfx950).
//
// Replaces the per-query SQL fetch + Python grouping + scoring loop of the reference
// (indexer/bm25_indexer.py:434-481).  The dense doc index is cut into tiles of TILE documents; one
// workgroup owns one (tile, query) pair and keeps the tile's float64 accumulators in LDS.  It first locates
// the tile's slice of every query term's posting list (two loads from the skip table for long lists, a
// wave-wide 64-ary search for short ones; one term per wave side by side), then, IN QUERY ORDER, streams each slice with coalesced loads; a document occurs at most once per posting list
// (PRIMARY KEY (doc_id, term), :100-104), so the read-modify-write of acc[doc] needs no atomics, and a
// barrier between terms makes the float64 summation order equal to the reference's (:466-478).
// The arithmetic is written operation by operation as Python evaluates it and this file is compiled with
// -ffp-contract=off, so scores are bit-identical to the oracle.
//
// HBM traffic per query: 8 B per posting of the query's terms + 4 B per document (doc_len) + 12 B per
// candidate document (the (score, doc) list consumed by the top-k select).
#include "msr_common.h"
#include "msr_internal.h"

namespace {

constexpr int BM25_TILE = MSR_BM25_TILE;
constexpr int BM25_THREADS = 256;
constexpr int BM25_MAX_TERMS = 64;                            // MSR_MAX_QUERY_TERMS
constexpr uint64_t UNTOUCHED = 0x7FF8DEADBEEF0001ull;   // a quiet-NaN payload no computation produces

// First index in [s, e) with a[idx] >= target (e if none).  Executed by one full wave.
__device__ __forceinline__ int64_t wave_lower_bound(const int32_t* __restrict__ a, int64_t s, int64_t e,
                                                    int32_t target) {
    const int lane = threadIdx.x & 63;
    while (e - s > 64) {
        const int64_t len = e - s;
        const int64_t chunk = (len + 63) >> 6;
        int64_t idx = s + (int64_t)(lane + 1) * chunk - 1;
        if (idx > e - 1) idx = e - 1;
        const bool ge = a[idx] >= target;
        const unsigned long long m = __ballot(ge);
        if (m == 0) return e;
        const int f = __ffsll((long long)m) - 1;
        const int64_t idx_f = __shfl(idx, f);
        const int64_t idx_p = __shfl(idx, f > 0 ? f - 1 : 0);
        if (f > 0) s = idx_p + 1;
        e = idx_f;                                   // a[idx_f] >= target: the answer is in [s, idx_f]
        if (e <= s) return s;
    }
    const int64_t i = s + lane;
    const bool ge = i < e && a[i] >= target;
    const unsigned long long m = __ballot(ge);
    if (m == 0) return e;
    return s + (__ffsll((long long)m) - 1);
}

__global__ __launch_bounds__(BM25_THREADS) void bm25_taat_kernel(Bm25Index ix,
                                                                  const int32_t* __restrict__ q_term_off,
                                                                  const int32_t* __restrict__ q_terms,
                                                                  const int32_t* __restrict__ q_qtf,
                                                                  int q_first, double min_score,
                                                                  double* __restrict__ cand_score,
                                                                  int32_t* __restrict__ cand_doc,
                                                                  int32_t* __restrict__ cand_n) {
    __shared__ double acc[BM25_TILE];
    __shared__ int32_t dl[BM25_TILE];
    __shared__ int64_t slice[2 * BM25_MAX_TERMS];                // [term slot][begin, end) of the tile's postings
    const int tid = threadIdx.x;
    // grid = (queries, tiles): consecutive workgroups score the SAME tile for different queries, so the tile's document
    // lengths and the slices of the terms the queries share (the city term is in every query, search_api.py:155-166)
    // are served by the L2 after the first of them
    const int q = blockIdx.x;                        // row of `scores`
    const int tile = blockIdx.y;
    const int64_t lo = (int64_t)tile * BM25_TILE;
    const int64_t hi = lo + BM25_TILE < ix.n_docs ? lo + BM25_TILE : ix.n_docs;
    const int n = (int)(hi - lo);
    // the tile's document lengths: issued now, parked in registers while the posting slices are located (both are
    // chains of dependent loads; side by side their latencies overlap), written to LDS afterwards
    int32_t dl_reg[BM25_TILE / BM25_THREADS];
#pragma unroll
    for (int u = 0; u < BM25_TILE / BM25_THREADS; ++u) {
        const int i = tid + u * BM25_THREADS;
        dl_reg[u] = i < n ? ix.doc_len[lo + i] : 0;
    }
    const double k1 = ix.k1, b = ix.b, avgdl = ix.avgdl;
    const double k1p1 = k1 + 1.0;                    // self.k1 + 1
    const double omb = 1.0 - b;                      // 1 - self.b
    const int t0 = q_term_off[q_first + q];
    int t1 = q_term_off[q_first + q + 1];
    if (t1 - t0 > BM25_MAX_TERMS) t1 = t0 + BM25_MAX_TERMS;      // the host never sends more
    // Locate the tile's slice of every posting list first, one term per wave at a time: the searches are
    // chains of dependent loads, so running them for all terms side by side hides most of their latency.
    {
        const int wv = tid >> 6;
        for (int j = t0 + wv; j < t1; j += BM25_THREADS / 64) {
            const int32_t t = q_terms[j];
            int64_t ps = 0, pe = 0;
            if (t >= 0 && t < ix.n_terms) {
                const int64_t s = ix.term_off[t], e = ix.term_off[t + 1];
                if (e > s) {
                    const int h = ix.heavy_id ? ix.heavy_id[t] : -1;
                    if (h >= 0) {                            // long list: the slice comes from the skip table
                        const uint32_t* row = ix.tile_off + (int64_t)h * (ix.n_tiles + 1) + tile;
                        ps = s + row[0];
                        pe = s + row[1];
                    } else {
                        ps = wave_lower_bound(ix.post_doc, s, e, (int32_t)lo);
                        pe = wave_lower_bound(ix.post_doc, ps, e, (int32_t)hi);
                    }
                }
            }
            if ((tid & 63) == 0) { slice[2 * (j - t0)] = ps; slice[2 * (j - t0) + 1] = pe; }
        }
    }
#pragma unroll
    for (int u = 0; u < BM25_TILE / BM25_THREADS; ++u) {
        const int i = tid + u * BM25_THREADS;
        acc[i] = __longlong_as_double((long long)UNTOUCHED);
        dl[i] = dl_reg[u];
    }
    __syncthreads();
    // One posting: the reference's arithmetic, operation by operation (:472-478).
    auto apply = [&](int32_t pdoc, int32_t ptf, double idf, double qtf) {
        const int d = pdoc - (int32_t)lo;
        const double tf = (double)ptf;
        const double dlen = (double)dl[d];
        // tf_component = (tf * (k1 + 1)) / (tf + k1 * (1 - b + b * doc_length / avg_doc_length))
        const double comp = (tf * k1p1) / (tf + k1 * (omb + (b * dlen) / avgdl));
        // term_score = idf * tf_component * query_term_freq[term]; bm25_score += term_score
        const double c = (idf * comp) * qtf;
        const double a = acc[d];
        acc[d] = ((uint64_t)__double_as_longlong(a) == UNTOUCHED ? 0.0 : a) + c;
    };
    // The rest of a slice, U postings per thread and round: all loads of a round are issued before the first is used.
    auto stream = [&](int64_t from, int64_t pe, double idf, double qtf) {
        constexpr int U = 4;
        for (int64_t i0 = from + tid; i0 < pe; i0 += (int64_t)U * BM25_THREADS) {
            int32_t pd[U], ptf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + (int64_t)u * BM25_THREADS;
                pd[u] = i < pe ? ix.post_doc[i] : -1;
                ptf[u] = i < pe ? ix.post_tf[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (pd[u] >= 0) apply(pd[u], ptf[u], idf, qtf);
        }
    };
    // The first 256 postings of the first TPRE terms' slices (for most terms: the whole slice) are fetched side by
    // side BEFORE the ordered accumulation starts, together with the terms' idf and query frequency: otherwise every
    // term costs a round trip to memory between two barriers.  The accumulation itself stays IN QUERY ORDER with a
    // barrier between terms: the float64 sums must match the reference's (:466-478).
    constexpr int TPRE = 6;
    int32_t pd0[TPRE], ptf0[TPRE];
    double idf0[TPRE], qtf0[TPRE];
#pragma unroll
    for (int jj = 0; jj < TPRE; ++jj) {
        pd0[jj] = -1; ptf0[jj] = 0; idf0[jj] = 0.0; qtf0[jj] = 0.0;
        if (t0 + jj < t1) {
            const int64_t ps = slice[2 * jj], pe = slice[2 * jj + 1];
            if (pe > ps) {
                const int32_t t = q_terms[t0 + jj];
                idf0[jj] = (double)ix.idf[t];
                qtf0[jj] = (double)q_qtf[t0 + jj];
                const int64_t i = ps + tid;
                if (i < pe) { pd0[jj] = ix.post_doc[i]; ptf0[jj] = ix.post_tf[i]; }
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < TPRE; ++jj) {
        if (t0 + jj >= t1) break;                                // block-uniform
        const int64_t ps = slice[2 * jj], pe = slice[2 * jj + 1];
        if (pe <= ps) continue;                                  // block-uniform
        if (pd0[jj] >= 0) apply(pd0[jj], ptf0[jj], idf0[jj], qtf0[jj]);
        stream(ps + BM25_THREADS, pe, idf0[jj], qtf0[jj]);
        __syncthreads();
    }
    for (int j = t0 + TPRE; j < t1; ++j) {
        const int64_t ps = slice[2 * (j - t0)], pe = slice[2 * (j - t0) + 1];
        if (pe <= ps) continue;                                  // block-uniform
        const int32_t t = q_terms[j];
        stream(ps, pe, (double)ix.idf[t], (double)q_qtf[j]);
        __syncthreads();
    }
    // Emit the tile's candidates (touched by a posting AND score >= min_score, :461,480) as (score, doc) pairs
    // appended to the query's list: one reservation per workgroup.  Most documents of a tile are not
    // candidates, so this replaces an 8 B/document dense row by 12 B per candidate.
    __shared__ int s_cnt, s_base;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n; i += BM25_THREADS) {
        const double a = acc[i];
        mine += ((uint64_t)__double_as_longlong(a) != UNTOUCHED && a >= min_score) ? 1 : 0;
    }
    int pos = mine ? atomicAdd(&s_cnt, mine) : 0;
    __syncthreads();
    if (tid == 0 && s_cnt) s_base = atomicAdd(&cand_n[q], s_cnt);
    __syncthreads();
    if (mine) {
        const int64_t o = (int64_t)q * ix.n_docs + s_base + pos;
        int w = 0;
        for (int i = tid; i < n; i += BM25_THREADS) {
            const double a = acc[i];
            if ((uint64_t)__double_as_longlong(a) != UNTOUCHED && a >= min_score) {
                cand_score[o + w] = a;
                cand_doc[o + w] = (int32_t)(lo + i);
                ++w;
            }
        }
    }
}

// Bind-time validation of the CSR the scoring kernel trusts: offsets monotone and complete, document indices
// in range and strictly ascending inside every posting list (that is what makes the LDS accumulation
// conflict-free and in-bounds), positive term frequencies, non-negative lengths.  flag starts at 0x7F7F7F7F and
// receives the lowest number among the violated rules.
__global__ __launch_bounds__(256) void validate_postings_kernel(Bm25Index ix, int32_t* __restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = g; t <= ix.n_terms; t += stride) {
        const int64_t o = ix.term_off[t];
        if ((t == 0 && o != 0) || (t == ix.n_terms && o != ix.n_postings) || (t < ix.n_terms && ix.term_off[t + 1] < o))
            atomicMin(flag, 1);
    }
    for (int64_t d = g; d < ix.n_docs; d += stride)
        if (ix.doc_len[d] < 0) atomicMin(flag, 4);
    for (int64_t i = g; i < ix.n_postings; i += stride) {
        const int32_t d = ix.post_doc[i];
        if (d < 0 || d >= ix.n_docs) { atomicMin(flag, 2); continue; }
        if (ix.post_tf[i] <= 0) atomicMin(flag, 5);
        if (i > 0 && d <= ix.post_doc[i - 1]) {
            // a descent is only legal at the first posting of a term: i must be one of the offsets
            int64_t lo = 0, hi = ix.n_terms;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (ix.term_off[mid] < i) lo = mid + 1; else hi = mid;
            }
            if (ix.term_off[lo] != i) atomicMin(flag, 3);
        }
    }
}

// One workgroup per heavy term: tile_off[h][j] = number of postings of the term with document < j * TILE.
__global__ __launch_bounds__(256) void build_skip_kernel(Bm25Index ix, const int32_t* __restrict__ heavy_terms,
                                                          uint32_t* __restrict__ tile_off) {
    const int h = blockIdx.x;
    const int32_t t = heavy_terms[h];
    const int64_t s = ix.term_off[t], e = ix.term_off[t + 1];
    for (int j = threadIdx.x; j <= ix.n_tiles; j += 256) {
        const int64_t target = (int64_t)j * BM25_TILE;
        int64_t lo = s, hi = e;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ix.post_doc[mid] < target) lo = mid + 1; else hi = mid;
        }
        tile_off[(int64_t)h * (ix.n_tiles + 1) + j] = (uint32_t)(lo - s);
    }
}

}  // namespace

hipError_t msr_bm25_build_skip(const Bm25Index& ix, const int32_t* heavy_terms, int n_heavy, uint32_t* tile_off,
                               hipStream_t stream) {
    if (n_heavy <= 0) return hipSuccess;
    build_skip_kernel<<<n_heavy, 256, 0, stream>>>(ix, heavy_terms, tile_off);
    return hipGetLastError();
}

hipError_t msr_bm25_validate(const Bm25Index& ix, int32_t* flag, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(flag, 0x7F, sizeof(int32_t), stream);
    if (err != hipSuccess) return err;
    validate_postings_kernel<<<2048, 256, 0, stream>>>(ix, flag);
    return hipGetLastError();
}

hipError_t msr_bm25_scores(const Bm25Index& ix, const int32_t* q_term_off, const int32_t* q_terms,
                           const int32_t* q_qtf, int q_first, int nq, double min_score, double* cand_score,
                           int32_t* cand_doc, int32_t* cand_n, hipStream_t stream) {
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    dim3 grid((unsigned)nq, (unsigned)((ix.n_docs + BM25_TILE - 1) / BM25_TILE));
    bm25_taat_kernel<<<grid, BM25_THREADS, 0, stream>>>(ix, q_term_off, q_terms, q_qtf, q_first, min_score, cand_score,
                                                        cand_doc, cand_n);
    return hipGetLastError();
}
