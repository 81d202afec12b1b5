// K1 -- BM25 term-at-a-time scoring over HBM-resident CSR postings (gfx950).
//
// Replaces the per-query SQL fetch + Python grouping + scoring loop of the reference
// (indexer/bm25_indexer.py:434-481).  The dense doc index is cut into tiles of TILE documents.  A workgroup owns a span
// of up to 8 consecutive tiles and FOUR queries: its four waves score one query each, every wave with its own float64
// accumulators in LDS, sharing the current tile's length norms k1 (1 - b + b dl / avgdl) (evaluated once per document
// at bind time, bm25_dnorm_kernel; staged in LDS per tile).  A wave
//   1. builds its query's PLAN once for all tiles of the workgroup, lane-parallel (lane j = term j): offsets, idf, query
//      frequency; for long lists the skip-table row segment (where each of the workgroup's tiles starts inside the list);
//      for lists of <= 64 postings the whole list; for the lists in between ONE round of 64 probes (a covering range for
//      the workgroup's document span; postings outside a tile are masked when applied);
//   2. per tile: derives every term's slice from the plan (no memory access) -- one tile AHEAD of the tile it is
//      accumulating (two register sets: the kernel is latency-bound, a wave hides its own loads behind its arithmetic),
//   3. fetches the first PFC x 64 postings of the first TPRE slices side by side, then accumulates IN QUERY ORDER.  A
//      wave's LDS operations execute in order, so the float64 summation order of every document equals the reference's
//      (:466-478) WITHOUT a barrier between terms; a document occurs at most once per posting list (PRIMARY KEY
//      (doc_id, term), :100-104), so the read-modify-write of acc[doc] needs no atomics,
//   4. appends the touched documents with score >= min_score to the query's candidate list (ballot prefix, one
//      reservation per wave and tile).
// The only workgroup barriers (two per tile) publish the length norms.  The arithmetic is written operation by
// operation as Python evaluates it and this file is compiled with -ffp-contract=off, so scores are bit-identical to
// the oracle.
//
// HBM traffic per query: 8 B per posting of the query's terms + 1 B per document (doc_len, shared by four queries) +
// 12 B per candidate document (the (score, doc) list consumed by the top-k select).
#include <stdio.h>

#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"

namespace {

constexpr int BM25_TILE = MSR_BM25_TILE;
constexpr int BM25_THREADS = 256;
constexpr int BM25_QPW = BM25_THREADS / 64;                   // queries per workgroup (one per wave)
#ifndef BM25_WPE
#define BM25_WPE 3                                             // waves per SIMD the register budget is cut for
#endif
constexpr int BM25_TPW = 8;                                   // at most this many consecutive tiles per workgroup
constexpr int BM25_MAX_TERMS = 64;                            // MSR_MAX_QUERY_TERMS: one lane per term
constexpr uint64_t UNTOUCHED = 0x7FF8DEADBEEF0001ull;   // a quiet-NaN payload no computation produces

__device__ __forceinline__ int64_t lane_i64(int64_t v, int j) {          // v of lane j (j wave-uniform)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), j);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double lane_f64(double v, int j) {
    return __longlong_as_double(lane_i64(__double_as_longlong(v), j));
}

__global__ __launch_bounds__(BM25_THREADS) __attribute__((amdgpu_waves_per_eu(BM25_WPE, BM25_WPE))) void bm25_taat_kernel(Bm25Index ix,
                                                                  const int32_t* __restrict__ q_term_off,
                                                                  const int32_t* __restrict__ q_terms,
                                                                  const int32_t* __restrict__ q_qtf,
                                                                  int q_first, int nq, double min_score, int tpw,
                                                                  double* __restrict__ cand_score,
                                                                  int32_t* __restrict__ cand_doc,
                                                                  int32_t* __restrict__ cand_n, int dbg_arg) {
#ifdef MSR_DIAG
    const int dbg = dbg_arg;   // timing experiments (wrong results): 1 no division, 2 no accumulator update, 4 no streaming
                               // beyond the prefetch, 8 no prefetch, 16 no emission, 32 no term lookup
#else
    constexpr int dbg = 0;
#endif
    __shared__ double acc_all[BM25_QPW][BM25_TILE];
    __shared__ double dn[BM25_TILE];                         // k1 * (1 - b + b * doc_length / avg_doc_length)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid = (query groups, tile spans): consecutive workgroups score the SAME tiles for other queries, so the tiles' document
    // lengths and the slices of the terms the queries share (the city term is in every query, search_api.py:155-166)
    // are served by the L2 after the first of them
    const int q = blockIdx.x * BM25_QPW + wave;              // row of the candidate lists
    const bool live = q < nq;                                // (wave-uniform)
    const int tile0 = (int)blockIdx.y * tpw;                 // this workgroup's tiles: tile0 .. tile0 + n_my - 1
    const int n_my = tile0 + tpw <= ix.n_tiles ? tpw : ix.n_tiles - tile0;
    double* acc = acc_all[wave];
    const double k1p1 = ix.k1 + 1.0;                 // self.k1 + 1
    // ---- 1. the query's plan, ONCE for all tiles of the workgroup; lane j = term j ----
    int nt = 0;
    int64_t s_v = 0;                                 // long list: its first posting
    uint32_t off_v[BM25_TPW + 1];                    // long list: where each of the workgroup's tiles starts inside it
    int64_t r0_v = 0, r1_v = 0;                      // other lists: the postings to look at, for EVERY tile of the workgroup
    double idf_v = 0.0, qtf_v = 0.0;
    bool heavy = false, medium = false;
#pragma unroll
    for (int i = 0; i <= BM25_TPW; ++i) off_v[i] = 0;
    if (live && !(dbg & 32)) {
        const int t0 = q_term_off[q_first + q];
        nt = q_term_off[q_first + q + 1] - t0;
        if (nt > BM25_MAX_TERMS) nt = BM25_MAX_TERMS;        // the host never sends more
        if (lane < nt) {
            const int32_t t = q_terms[t0 + lane];
            if (t >= 0 && t < ix.n_terms) {
                const int64_t s = ix.term_off[t], e = ix.term_off[t + 1];
                if (e > s) {
                    idf_v = (double)ix.idf[t];
                    qtf_v = (double)q_qtf[t0 + lane];
                    const int h = ix.heavy_id ? ix.heavy_id[t] : -1;
                    if (h >= 0) {                            // long list: the slices come from the skip table, one row
                        heavy = true;                        // segment for all the workgroup's tiles
                        s_v = s;
                        const uint32_t* row = ix.tile_off + (int64_t)h * (ix.n_tiles + 1) + tile0;
#pragma unroll
                        for (int i = 0; i <= BM25_TPW; ++i) off_v[i] = row[i < n_my ? i : n_my];
                    } else {
                        r0_v = s;                            // short list: all of it (postings of other tiles are masked)
                        r1_v = e;
                        medium = e - s > 64;
                    }
                }
            }
        }
    }
    // ---- 2. lists of 65 .. HEAVY_DF-1 postings: one round of 64 probes narrows [r0, r1) to the chunks that can hold
    //         documents of the workgroup's tiles; MED lists side by side (the probes are independent loads) ----
    {
        const int64_t lo8 = (int64_t)tile0 * BM25_TILE;
        const int64_t hi8 = lo8 + (int64_t)n_my * BM25_TILE < ix.n_docs ? lo8 + (int64_t)n_my * BM25_TILE : ix.n_docs;
        constexpr int MED = 4;
        unsigned long long todo = __ballot(medium);
        while (todo) {
            int jj[MED];
            int64_t s_[MED], e_[MED], ch_[MED];
            int32_t probe[MED];
#pragma unroll
            for (int m = 0; m < MED; ++m) {
                jj[m] = todo ? __ffsll((long long)todo) - 1 : -1;
                if (todo) todo &= todo - 1;
                probe[m] = 0; s_[m] = e_[m] = ch_[m] = 0;
                if (jj[m] >= 0) {                            // wave-uniform
                    s_[m] = lane_i64(r0_v, jj[m]);
                    e_[m] = lane_i64(r1_v, jj[m]);
                    ch_[m] = (e_[m] - s_[m] + 63) >> 6;
                    int64_t idx = s_[m] + (int64_t)(lane + 1) * ch_[m] - 1;     // last posting of chunk `lane`
                    if (idx > e_[m] - 1) idx = e_[m] - 1;
                    probe[m] = ix.post_doc[idx];
                }
            }
#pragma unroll
            for (int m = 0; m < MED; ++m) {
                if (jj[m] < 0) continue;                     // wave-uniform
                const unsigned long long ge_lo = __ballot((int64_t)probe[m] >= lo8);
                const unsigned long long ge_hi = __ballot((int64_t)probe[m] >= hi8);
                int64_t ps = e_[m], pe = e_[m];              // nothing >= lo: empty
                if (ge_lo) {
                    ps = s_[m] + (int64_t)(__ffsll((long long)ge_lo) - 1) * ch_[m];
                    if (ps > e_[m]) ps = e_[m];
                    if (ge_hi) {
                        pe = s_[m] + (int64_t)(__ffsll((long long)ge_hi)) * ch_[m];
                        if (pe > e_[m]) pe = e_[m];
                    }
                }
                if (lane == jj[m]) { r0_v = ps; r1_v = pe; }
            }
        }
    }
    // Two tiles are in flight per wave: the loads of tile t + 1 (length norms, the first postings of every slice) are issued
    // BEFORE tile t is accumulated and emitted, into a second register set -- the pass is latency-bound (its time follows
    // the number of resident waves), so every wave hides its own memory latency behind its own arithmetic.
    constexpr int TPRE = 5, PFC = 3;
    constexpr int DPT2 = BM25_TILE / BM25_THREADS / 2;       // 16-byte pieces of the length norms per thread
    struct TileRegs {
        int32_t pd[TPRE][PFC], tf[TPRE][PFC];                // the first PFC x 64 postings of the first TPRE slices
        double2 dn[DPT2];                                    // length norms: thread t holds documents 2 DPT2 t .. of the tile
        int64_t ps, pe;                                      // lane j: this tile's slice of term j
    };
    auto issue = [&](int tt, TileRegs& r) {
        const int64_t lo = (int64_t)(tile0 + tt) * BM25_TILE;
#pragma unroll
        for (int u = 0; u < DPT2; ++u) r.dn[u] = ((const double2*)(ix.dnorm + lo))[DPT2 * tid + u];      // (padded to whole tiles)
        // the tile's slice of every list (no memory access: the plan holds everything)
        r.ps = r0_v; r.pe = r1_v;
        if (heavy) {
            uint32_t o0 = 0, o1 = 0;
#pragma unroll
            for (int i = 0; i < BM25_TPW; ++i)
                if (i == tt) { o0 = off_v[i]; o1 = off_v[i + 1]; }
            r.ps = s_v + o0;
            r.pe = s_v + o1;
        }
        // the first PFC x 64 postings of the first TPRE slices (for nearly all terms: the whole slice): one memory round
        // trip for the postings of all terms
#pragma unroll
        for (int j = 0; j < TPRE; ++j) {
#pragma unroll
            for (int c = 0; c < PFC; ++c) { r.pd[j][c] = -1; r.tf[j][c] = 0; }
            if (j < nt && !(dbg & 8)) {                          // wave-uniform
                const int64_t ps = lane_i64(r.ps, j), pe = lane_i64(r.pe, j);
#pragma unroll
                for (int c = 0; c < PFC; ++c) {
                    if (ps + 64 * c < pe) {                      // wave-uniform: no instruction for chunks past the slice
                        const int64_t i = ps + 64 * c + lane;
                        if (i < pe) { const int2 p = ix.post[i]; r.pd[j][c] = p.x; r.tf[j][c] = p.y; }
                    }
                }
            }
        }
    };
    auto process = [&](int tt, TileRegs& r) {
        const int tile = tile0 + tt;
        const int64_t lo = (int64_t)tile * BM25_TILE;
        const int64_t hi = lo + BM25_TILE < ix.n_docs ? lo + BM25_TILE : ix.n_docs;
        const int n = (int)(hi - lo);
        const int64_t ps_v = r.ps, pe_v = r.pe;
        // accumulators of this wave's query; the length norms of the tile (shared by the four waves)
        {
            const double un = __longlong_as_double((long long)UNTOUCHED);
#pragma unroll
            for (int u = 0; u < BM25_TILE / 128; ++u) ((double2*)acc)[lane + 64 * u] = make_double2(un, un);
        }
        __syncthreads();                                         // every wave is done with the previous tile's norms
#pragma unroll
        for (int u = 0; u < DPT2; ++u) ((double2*)dn)[DPT2 * tid + u] = r.dn[u];
        __syncthreads();
        // U postings of ONE term per lane: the reference's arithmetic, operation by operation (:472-478), written so that the
        // U chains (a float64 division is 11 dependent instructions) are independent and interleave: nothing is branched
        // around -- a posting of another tile (covering ranges) or a missing one computes on document 0 and only its final
        // store is masked.  A document occurs once per posting list, so the U read-modify-writes never touch the same slot.
        auto apply = [&](auto u_c, const int32_t* pdoc, const int32_t* ptf, double idf, double qtf) {
            constexpr int U = decltype(u_c)::value;
            bool ok[U];
            uint32_t d[U];
            double c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t dd = (uint32_t)(pdoc[u] - (int32_t)lo);
                ok[u] = dd < (uint32_t)n;
                d[u] = ok[u] ? dd : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double tf = (double)ptf[u];
                // tf_component = (tf * (k1 + 1)) / (tf + k1 * (1 - b + b * doc_length / avg_doc_length))
                const double comp = (dbg & 1) ? (tf * k1p1) * (tf + dn[d[u]]) : (tf * k1p1) / (tf + dn[d[u]]);
                // term_score = idf * tf_component * query_term_freq[term]; bm25_score += term_score
                c[u] = (idf * comp) * qtf;
            }
            if (dbg & 2) return;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double a = acc[d[u]];
                const double x = ((uint64_t)__double_as_longlong(a) == UNTOUCHED ? 0.0 : a) + c[u];
                if (ok[u]) acc[d[u]] = x;
            }
        };
        // The rest of a slice, U x 64 postings per round: all loads of a round are issued before the first is used.
        auto stream = [&](int64_t from, int64_t pe, double idf, double qtf) {
            constexpr int U = 4;
            if (dbg & 4) return;
            for (int64_t base = from; base < pe; base += (int64_t)U * 64) {
                int32_t pd[U], ptf[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t i = base + lane + (int64_t)u * 64;
                    const int2 p = i < pe ? ix.post[i] : make_int2(-1, 0);
                    pd[u] = p.x;
                    ptf[u] = p.y;
                }
                apply(std::integral_constant<int, U>{}, pd, ptf, idf, qtf);
            }
        };
#pragma unroll
        for (int j = 0; j < TPRE; ++j) {
            if (j >= nt) break;                                  // wave-uniform
            const int64_t ps = lane_i64(ps_v, j), pe = lane_i64(pe_v, j);
            if (pe <= ps) continue;
            const double idf = lane_f64(idf_v, j), qtf = lane_f64(qtf_v, j);
            apply(std::integral_constant<int, PFC>{}, r.pd[j], r.tf[j], idf, qtf);     // (chunks past the slice hold doc -1)
            if (pe - ps > 64 * PFC) stream(ps + 64 * PFC, pe, idf, qtf);
        }
        for (int j = TPRE; j < nt; ++j) {
            const int64_t ps = lane_i64(ps_v, j), pe = lane_i64(pe_v, j);
            if (pe <= ps) continue;
            stream(ps, pe, lane_f64(idf_v, j), lane_f64(qtf_v, j));
        }
        if (!live || (dbg & 16)) return;                       // (wave-uniform; the barriers are at the top of the loop)
        // ---- 4. the tile's candidates (touched by a posting AND score >= min_score, :461,480) as (score, doc) pairs
        //         appended to the query's list: one reservation per wave.  Most documents of a tile are not candidates, so
        //         this replaces an 8 B/document dense row by 12 B per candidate. ----
        // lane l looks at documents 2 l + 128 u and 2 l + 128 u + 1 (one 16-byte LDS read per round)
        int total = 0;
        unsigned long long f0[BM25_TILE / 128], f1[BM25_TILE / 128];
#pragma unroll
        for (int u = 0; u < BM25_TILE / 128; ++u) {
            const int i = 2 * lane + 128 * u;
            const double2 a = ((const double2*)acc)[lane + 64 * u];
            f0[u] = __ballot(i < n && (uint64_t)__double_as_longlong(a.x) != UNTOUCHED && a.x >= min_score);
            f1[u] = __ballot(i + 1 < n && (uint64_t)__double_as_longlong(a.y) != UNTOUCHED && a.y >= min_score);
            total += __popcll(f0[u]) + __popcll(f1[u]);
        }
        if (total == 0) return;                                // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&cand_n[q], total);
        base = __builtin_amdgcn_readfirstlane(base);
        const int64_t o = (int64_t)q * ix.n_docs + base;
        int run = 0;
        const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
        for (int u = 0; u < BM25_TILE / 128; ++u) {
            const int i = 2 * lane + 128 * u;
            if ((f0[u] >> lane) & 1) {
                const int w = run + __popcll(f0[u] & below);
                cand_score[o + w] = acc[i];
                cand_doc[o + w] = (int32_t)(lo + i);
            }
            run += __popcll(f0[u]);
            if ((f1[u] >> lane) & 1) {
                const int w = run + __popcll(f1[u] & below);
                cand_score[o + w] = acc[i + 1];
                cand_doc[o + w] = (int32_t)(lo + i + 1);
            }
            run += __popcll(f1[u]);
        }
    };
    TileRegs ra, rb;
    issue(0, ra);
    for (int tt = 0; tt < n_my; tt += 2) {
        if (tt + 1 < n_my) issue(tt + 1, rb);
        process(tt, ra);
        if (tt + 1 < n_my) {
            if (tt + 2 < n_my) issue(tt + 2, ra);
            process(tt + 1, rb);
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

// post[i] = {post_doc[i], post_tf[i]}: the scoring kernel reads a posting with ONE 8-byte load (the engine-owned copy
// costs 8 B per posting of HBM; with two 4-byte arrays the kernel issued twice the memory instructions, and their issue
// rate -- not bandwidth -- is what the per-phase wave clocks showed to matter)
__global__ __launch_bounds__(256) void interleave_postings_kernel(const int32_t* __restrict__ post_doc,
                                                                   const int32_t* __restrict__ post_tf, int64_t n,
                                                                   int2* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = make_int2(post_doc[i], post_tf[i]);
}

// dnorm[d] = k1 * (1 - b + b * doc_length / avg_doc_length) for every document, padded to whole tiles (1.0 beyond n_docs):
// the scoring kernel's length norm, evaluated ONCE with the expression the reference evaluates per posting (:474)
__global__ __launch_bounds__(256) void bm25_dnorm_kernel(const int32_t* __restrict__ doc_len, int64_t n_docs, int64_t n_pad,
                                                          double k1, double b, double avgdl, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    const double omb = 1.0 - b;
    out[i] = i < n_docs ? k1 * (omb + (b * (double)doc_len[i]) / avgdl) : 1.0;
}

int g_bm25_dbg = 0;

}  // namespace

void msr_bm25_set_dbg(int v) { g_bm25_dbg = v; }      // honoured by -DMSR_DIAG builds only
hipError_t msr_bm25_build_skip(const Bm25Index& ix, const int32_t* heavy_terms, int n_heavy, uint32_t* tile_off,
                               hipStream_t stream) {
    if (n_heavy <= 0) return hipSuccess;
    build_skip_kernel<<<n_heavy, 256, 0, stream>>>(ix, heavy_terms, tile_off);
    return hipGetLastError();
}

hipError_t msr_bm25_interleave(const int32_t* post_doc, const int32_t* post_tf, int64_t n, void* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    interleave_postings_kernel<<<4096, 256, 0, stream>>>(post_doc, post_tf, n, (int2*)out);
    return hipGetLastError();
}

hipError_t msr_bm25_dnorm(const int32_t* doc_len, int64_t n_docs, int64_t n_pad, double k1, double b, double avgdl, double* out,
                          hipStream_t stream) {
    if (n_pad <= 0) return hipSuccess;
    bm25_dnorm_kernel<<<(unsigned)((n_pad + 255) / 256), 256, 0, stream>>>(doc_len, n_docs, n_pad, k1, b, avgdl, out);
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
    // A workgroup looks a query's terms up once and then walks `tpw` consecutive tiles with that plan in registers.  More tiles
    // per workgroup amortise the lookups (chains of dependent loads) but leave fewer workgroups: large batches take 8, a
    // single query keeps one tile per workgroup (its ~1000 waves are all the parallelism it has).
    const int groups = (nq + BM25_QPW - 1) / BM25_QPW;
    int tpw = groups >= 16 ? 8 : groups >= 8 ? 4 : groups >= 4 ? 2 : 1;
    while (tpw > 1 && (int64_t)groups * ((ix.n_tiles + tpw - 1) / tpw) < 2048) tpw >>= 1;
    dim3 grid((unsigned)groups, (unsigned)((ix.n_tiles + tpw - 1) / tpw));
    // (diagnostic build: bit 6 of the knock-out mask pads every workgroup with 41 KB of unused LDS -> 2 instead of 4 per CU)
    const size_t pad_lds = (g_bm25_dbg & 64) ? 41 * 1024 : 0;
    bm25_taat_kernel<<<grid, BM25_THREADS, pad_lds, stream>>>(ix, q_term_off, q_terms, q_qtf, q_first, nq, min_score, tpw, cand_score,
                                                        cand_doc, cand_n, g_bm25_dbg);
    return hipGetLastError();
}
