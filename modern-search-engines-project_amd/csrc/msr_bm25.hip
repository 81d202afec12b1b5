// K1 -- BM25 term-at-a-time scoring over HBM-resident CSR postings (gfx950).
//
// Replaces the per-query SQL fetch + Python grouping + scoring loop of the reference
// (indexer/bm25_indexer.py:434-481).  The dense doc index is cut into tiles of TILE documents; a WAVE scores one query
// over a span of up to 8 consecutive tiles, with its own float64 accumulators (one per document of the current tile) and
// the list of the documents it has touched in LDS.  Waves are independent: no workgroup barrier anywhere.
//
// What is streamed and what is looked up.  A posting is read as {doc, tf_component} (12 bytes): the tf_component
// (tf (k1 + 1)) / (tf + k1 (1 - b + b dl / avgdl)) (:473-475) depends on (tf, document) only and is evaluated ONCE at bind
// time with the reference's own operations (bm25_post_comp_kernel), so the kernel neither divides nor looks a length up.
// A term whose idf is NEGATIVE (document frequency above half the corpus: the city term that search_api.py:160-164 puts
// into every query) can only lower a score: every one of its contributions (idf * tf_component) * qtf is < 0.  With
// min_score >= 0 (:480) a document matched by such terms alone is never a candidate, so their lists -- 96 % of the
// postings a benchmark query names -- are not streamed at all: the candidates are the documents touched by the OTHER terms
// ("streamed" terms), and a negative term's contribution to one of them is looked up in a dense per-document table of its
// tf_components (built at bind for the long negative lists, 8 B per document).  With min_score < 0, or for a negative
// term without a table, every list is streamed; the result is the same either way, bit for bit.
//
// Summation order.  The reference adds the contributions of a document's terms in the query's first-occurrence order
// (:466-478) in float64.  A wave walks its query's terms in that order:
//   streamed term j: for each posting of the tile, acc[doc] += (idf * tfc) * qtf as an LDS read-modify-write (a wave's LDS
//     operations execute in order; a document occurs at most once per list, PRIMARY KEY (doc_id, term) :100-104, so no
//     atomics).  A document touched for the FIRST time starts from 0.0 plus the looked-up contributions of the negative
//     terms BEFORE j, in order, and is appended to the wave's list;
//   looked-up term i: for each document on the list so far, acc[doc] += (idf * table_i[doc]) * qtf if the table has it.
// So every candidate's sum is the reference's sum operation by operation; this file is compiled with -ffp-contract=off.
// At the end of a tile the list is what is emitted (score >= min_score, :480) and what is reset -- there is no pass over
// all TILE accumulators.
//
// The plan (where each term's slices are: skip-table row segment for long lists, the whole list for <= 64 postings, ONE
// round of 64 probes for the lists in between) is built once per span, lane-parallel (lane j = term j); the loads of
// tile t + 1 are issued before tile t is accumulated (two register sets).
//
// Output: the candidates of a (query, span) go to a segment of the query's candidate row that belongs to that wave alone
// -- no atomics, no counters to clear; the top-k select (msr_topk.hip) walks the segments.
//
// HBM traffic per query: 12 B per posting of its streamed terms + 8 B per (touched document, looked-up term) from tables
// that stay in the L2 / Infinity Cache + 12 B per candidate (the (score, doc) list consumed by the top-k select).
#include <stdio.h>

#include <type_traits>

#include "msr_common.h"
#include "msr_internal.h"

namespace {

constexpr int BM25_TILE = MSR_BM25_TILE;
#ifndef BM25_WPW
#define BM25_WPW 1                                             // one-wave workgroups: work items differ by an order of magnitude (a long
#endif                                                         // positive list or not) and a workgroup holds its LDS until its slowest
constexpr int BM25_WAVES = BM25_WPW;                          // wave is done (measured: 1 wave 0.18 ms, 2 waves 0.21, 4 waves 0.23)
constexpr int BM25_THREADS = 64 * BM25_WAVES;
constexpr int BM25_TPW = 8;                                   // at most this many consecutive tiles per work item
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
__device__ __forceinline__ int lane_rank(unsigned long long m) {         // number of set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// one posting = one 12-byte load
__device__ __forceinline__ double post_comp(const Bm25Post& x) { return __hiloint2double((int)x.comp_hi, (int)x.comp_lo); }
// a table row as a GLOBAL pointer: rebuilt from the integer a lane carries it would be a flat pointer, and flat loads return
// out of order -- every one of them makes the wave wait for ALL its outstanding memory operations (vmcnt(0) + lgkmcnt(0))
typedef const __attribute__((address_space(1))) double* gtable;
__device__ __forceinline__ gtable table_of(int64_t bits) { return (gtable)(uint64_t)bits; }

enum : int { K_DEAD = 0, K_LOOKUP = 1, K_HEAVY = 2, K_RANGE = 3 };

#ifndef BM25_WPE
#define BM25_WPE 4                                             // waves per SIMD the register budget is cut for
#endif
__global__ __launch_bounds__(BM25_THREADS) __attribute__((amdgpu_waves_per_eu(BM25_WPE, BM25_WPE))) void bm25_taat_kernel(Bm25Index ix,
                                                                  const int32_t* __restrict__ q_term_off,
                                                                  const int32_t* __restrict__ q_terms,
                                                                  const int32_t* __restrict__ q_qtf,
                                                                  int q_first, int nq, double min_score, int tpw, int n_spans,
                                                                  double* __restrict__ cand_score,
                                                                  int32_t* __restrict__ cand_doc,
                                                                  int32_t* __restrict__ seg_n, int dbg_arg) {
#ifdef MSR_DIAG
    const int dbg = dbg_arg;   // timing experiments (wrong results): 1 no table lookups, 2 no accumulator update, 4 no streaming
                               // beyond the prefetch, 8 no prefetch, 16 no emission, 64 stream every list (no pruning)
#else
    constexpr int dbg = 0;
#endif
    __shared__ double acc_all[BM25_WAVES][BM25_TILE];
    __shared__ uint16_t list_all[BM25_WAVES][BM25_TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work item = (query, span of tiles), the query running fastest: the waves of a workgroup and of its neighbours score the
    // SAME tiles for different queries, so the table lines and the slices of terms the queries share come from the L2
    const int item = (int)blockIdx.x * BM25_WAVES + wave;
    const int span = item / nq, q = item - span * nq;          // q: row of the candidate lists
    if (span >= n_spans) return;                               // (wave-uniform; there is no barrier in this kernel)
    // The candidates of this item go to segment `span` of row q of the candidate arrays: positions [span tpw TILE, ...) of
    // the row belong to this wave alone (a span cannot yield more candidates than it has documents), so nothing is reserved
    // with an atomic; the segment's length is written once, at the end.
    int32_t* seg_len = seg_n + (int64_t)q * n_spans + span;
    int seg_cnt = 0;
    const int tile0 = span * tpw;                              // this wave's tiles: tile0 .. tile0 + n_my - 1
    const int n_my = tile0 + tpw <= ix.n_tiles ? tpw : ix.n_tiles - tile0;
    double* acc = acc_all[wave];
    uint16_t* list = list_all[wave];
    {   // every accumulator starts untouched; a tile resets exactly the ones it touched
        const double un = __longlong_as_double((long long)UNTOUCHED);
#pragma unroll
        for (int u = 0; u < BM25_TILE / 128; ++u) ((double2*)acc)[lane + 64 * u] = make_double2(un, un);
    }
    // a negative term's list need not be streamed when nothing below 0 can be a candidate (:480)
    const bool prune = min_score >= 0.0 && ix.dense_id != nullptr && !(dbg & 64);
    // ---- 1. the query's plan, ONCE for all tiles of the span; lane j = term j ----
    int nt = 0, klass = K_DEAD;
    int64_t s_v = 0;                                 // long list: its first posting; looked-up term: its table row
    uint32_t off_v[BM25_TPW + 1];                    // long list: where each of the span's tiles starts inside it
    int64_t r0_v = 0, r1_v = 0;                      // other lists: the postings to look at, for EVERY tile of the span
    double idf_v = 0.0, qtf_v = 0.0;
    bool medium = false;
#pragma unroll
    for (int i = 0; i <= BM25_TPW; ++i) off_v[i] = 0;
    {
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
                    const int dh = prune ? ix.dense_id[t] : -1;
                    const int h = ix.heavy_id ? ix.heavy_id[t] : -1;
                    if (dh >= 0 && idf_v < 0.0 && qtf_v > 0.0) {   // every contribution < 0: looked up, never streamed
                        klass = K_LOOKUP;
                        s_v = (int64_t)(ix.dense_comp + (int64_t)dh * ix.dense_stride);
                    } else if (h >= 0) {                     // long list: the slices come from the skip table, one row
                        klass = K_HEAVY;                     // segment for all the span's tiles
                        s_v = s;
                        const uint32_t* row = ix.tile_off + (int64_t)h * (ix.n_tiles + 1) + tile0;
#pragma unroll
                        for (int i = 0; i <= BM25_TPW; ++i) off_v[i] = row[i < n_my ? i : n_my];
                    } else {
                        klass = K_RANGE;
                        r0_v = s;                            // short list: all of it (postings of other tiles are masked)
                        r1_v = e;
                        medium = e - s > 64;
                    }
                }
            }
        }
    }
    const unsigned long long look_mask = __ballot(klass == K_LOOKUP);
    const unsigned long long stream_mask = __ballot(klass >= K_HEAVY);
    if (stream_mask == 0) {                          // no streamed term: no document can reach min_score (or no known term)
        if (lane == 0) *seg_len = 0;
        return;
    }
    const int64_t seg_base = (int64_t)q * ix.n_docs + (int64_t)tile0 * BM25_TILE;
    // ---- 2. lists of 65 .. HEAVY_DF-1 postings: one round of 64 probes narrows [r0, r1) to the chunks that can hold
    //         documents of the span's tiles; MED lists side by side (the probes are independent loads) ----
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
                    probe[m] = ix.post[idx].doc;
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
    // the first TPRE streamed terms get their first PFC x 64 postings of a tile prefetched (for nearly all terms: the whole slice)
    constexpr int TPRE = 4, PFC = 2;
    int sj[TPRE];
    {
        unsigned long long m = stream_mask;
#pragma unroll
        for (int k = 0; k < TPRE; ++k) {
            sj[k] = m ? __ffsll((long long)m) - 1 : -1;
            if (m) m &= m - 1;
        }
    }
    const int j_rest = sj[TPRE - 1] >= 0 ? sj[TPRE - 1] + 1 : nt;       // terms from here on are not prefetched
    // Two tiles are in flight per wave: the loads of tile t + 1 are issued BEFORE tile t is accumulated and emitted, into a
    // second register set -- the pass is latency-bound, every wave hides its own memory latency behind its own arithmetic.
    // Inside a chunk a lane without a posting loads the sentinel {doc -1, 0.0} the bind step put behind the last posting: no
    // load sits in a divergent branch, and the loaded words are not touched before they are used.
    const int64_t null_post = ix.n_postings;
    struct TileRegs {
        Bm25Post p[TPRE][PFC];                               // the first PFC x 64 postings of the first TPRE streamed slices
        int64_t ps, pe;                                      // lane j: this tile's slice of term j
    };
    auto issue = [&](int tt, TileRegs& r) {
        // the tile's slice of every list (no memory access: the plan holds everything)
        r.ps = r0_v; r.pe = r1_v;
        if (klass == K_HEAVY) {
            uint32_t o0 = 0, o1 = 0;
#pragma unroll
            for (int i = 0; i < BM25_TPW; ++i)
                if (i == tt) { o0 = off_v[i]; o1 = off_v[i + 1]; }
            r.ps = s_v + o0;
            r.pe = s_v + o1;
        }
#pragma unroll
        for (int k = 0; k < TPRE; ++k) {
#pragma unroll
            for (int c = 0; c < PFC; ++c) r.p[k][c].doc = -1;
            if (sj[k] >= 0 && !(dbg & 8)) {                  // wave-uniform
                const int64_t ps = lane_i64(r.ps, sj[k]), pe = lane_i64(r.pe, sj[k]);
#pragma unroll
                for (int c = 0; c < PFC; ++c) {
                    if (ps + 64 * c < pe) {                  // wave-uniform: no instruction for chunks past the slice
                        const int64_t i = ps + 64 * c + lane;
                        r.p[k][c] = ix.post[i < pe ? i : null_post];
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
        int list_n = 0;                                      // documents touched so far in this tile (wave-uniform)
        // U postings of streamed term j per lane.  Nothing is branched around per lane: a posting of another tile (covering
        // ranges) or a missing one computes on document 0 and only its stores are masked.  A document occurs once per
        // posting list, so the lanes of one step never touch the same slot.
        auto apply = [&](auto u_c, const int32_t* pdoc, const double* pcomp, double idf, double qtf, int j) {
            constexpr int U = decltype(u_c)::value;
            bool ok[U];
            uint32_t d[U];
            double c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t dd = (uint32_t)(pdoc[u] - (int32_t)lo);
                ok[u] = dd < (uint32_t)n;
                d[u] = ok[u] ? dd : 0u;
                // term_score = idf * tf_component * query_term_freq[term]  (:478)
                c[u] = (idf * pcomp[u]) * qtf;
            }
            if (dbg & 2) return;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double a = acc[d[u]];
                const bool first = ok[u] && (uint64_t)__double_as_longlong(a) == UNTOUCHED;
                const unsigned long long fm = __ballot(first);
                if (fm) {                                    // (wave-uniform) first touch: bm25_score = 0.0 (:466), then the
                    double a0 = 0.0;                         // negative terms that precede term j in the query, in order
                    unsigned long long lm = (dbg & 1) ? 0ull : look_mask & ((1ull << j) - 1ull);
                    while (lm) {
                        const int i = __ffsll((long long)lm) - 1;
                        lm &= lm - 1;
                        const gtable tbl = table_of(lane_i64(s_v, i)) + lo;
                        const double tc = first ? tbl[d[u]] : 0.0;
                        const double ci = (lane_f64(idf_v, i) * tc) * lane_f64(qtf_v, i);
                        if (tc != 0.0) a0 = a0 + ci;         // (0.0: the document lacks term i)
                    }
                    if (first) { list[list_n + lane_rank(fm)] = (uint16_t)d[u]; a = a0; }
                    list_n += __popcll(fm);
                }
                const double x = a + c[u];                   // bm25_score += term_score
                if (ok[u]) acc[d[u]] = x;
            }
        };
        // The rest of a slice, U x 64 postings per round: all loads of a round are issued before the first is used.
        auto stream = [&](int64_t from, int64_t pe, double idf, double qtf, int j) {
            constexpr int U = 4;
            if (dbg & 4) return;
            for (int64_t base = from; base < pe; base += (int64_t)U * 64) {
                int32_t pd[U];
                double pc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t i = base + lane + (int64_t)u * 64;
                    const Bm25Post x = ix.post[i < pe ? i : null_post];
                    pd[u] = x.doc; pc[u] = post_comp(x);
                }
                apply(std::integral_constant<int, U>{}, pd, pc, idf, qtf, j);
            }
        };
        // looked-up term i: its contribution to every document touched so far
        auto walk = [&](int i) {
            if (list_n == 0 || (dbg & 1)) return;
            const gtable tbl = table_of(lane_i64(s_v, i)) + lo;
            const double idf = lane_f64(idf_v, i), qtf = lane_f64(qtf_v, i);
            for (int b = 0; b < list_n; b += 128) {          // two batches of gathers in flight
                const int e0 = b + lane, e1 = b + 64 + lane;
                const uint32_t d0 = list[e0 < list_n ? e0 : 0], d1 = list[e1 < list_n ? e1 : 0];
                const double t0 = e0 < list_n ? tbl[d0] : 0.0, t1 = e1 < list_n ? tbl[d1] : 0.0;
                if (t0 != 0.0) acc[d0] = acc[d0] + (idf * t0) * qtf;
                if (t1 != 0.0) acc[d1] = acc[d1] + (idf * t1) * qtf;
            }
        };
        auto walks = [&](int from, int to) {                 // the looked-up terms at positions [from, to)
            unsigned long long lm = look_mask & ~((1ull << from) - 1ull);
            if (to < 64) lm &= (1ull << to) - 1ull;
            while (lm) {
                const int i = __ffsll((long long)lm) - 1;
                lm &= lm - 1;
                walk(i);
            }
        };
        // ---- the terms in query order: the first TPRE streamed ones from the prefetched registers ----
        int jprev = 0;
#pragma unroll
        for (int k = 0; k < TPRE; ++k) {
            const int j = sj[k];
            if (j < 0) break;                                // wave-uniform
            walks(jprev, j);
            jprev = j + 1;
            const int64_t ps = lane_i64(ps_v, j), pe = lane_i64(pe_v, j);
            if (pe <= ps) continue;
            const double idf = lane_f64(idf_v, j), qtf = lane_f64(qtf_v, j);
            int32_t pd[PFC];
            double pc[PFC];
#pragma unroll
            for (int c = 0; c < PFC; ++c) { pd[c] = r.p[k][c].doc; pc[c] = post_comp(r.p[k][c]); }   // (no posting: doc -1)
            if (pe - ps <= 64) apply(std::integral_constant<int, 1>{}, pd, pc, idf, qtf, j);
            else apply(std::integral_constant<int, PFC>{}, pd, pc, idf, qtf, j);
            if (pe - ps > 64 * PFC) stream(ps + 64 * PFC, pe, idf, qtf, j);
        }
        for (int j = j_rest; j < nt; ++j) {
            const int kj = __builtin_amdgcn_readlane(klass, j);
            if (kj == K_LOOKUP) { walk(j); continue; }
            if (kj == K_DEAD) continue;
            const int64_t ps = lane_i64(ps_v, j), pe = lane_i64(pe_v, j);
            if (pe <= ps) continue;
            stream(ps, pe, lane_f64(idf_v, j), lane_f64(qtf_v, j), j);
        }
        if (jprev < j_rest) walks(jprev, j_rest);            // (fewer than TPRE streamed terms: the looked-up ones behind the last)
        if (list_n == 0) return;
        // ---- the tile's candidates: the touched documents with score >= min_score (:461,480) as (score, doc) pairs appended
        //      to the item's segment; every touched accumulator goes back to "untouched" ----
        const double un = __longlong_as_double((long long)UNTOUCHED);
        for (int b = 0; b < list_n; b += 64) {
            const int e = b + lane;
            const bool in = e < list_n;
            const uint32_t d = list[in ? e : 0];
            const double sc = acc[d];
            const bool keep = in && sc >= min_score && !(dbg & 16);
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const int64_t w = seg_base + seg_cnt + lane_rank(km);
                cand_score[w] = sc;
                cand_doc[w] = (int32_t)(lo + d);
            }
            seg_cnt += __popcll(km);
            if (in) acc[d] = un;
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
    if (lane == 0) *seg_len = seg_cnt;
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

// post[i] = {post_doc[i], tf_component}: what the scoring kernel streams, one 12-byte load per posting; post[n] = {-1, 0.0}.
// tf_component = (tf * (k1 + 1)) / (tf + k1 * (1 - b + b * doc_length / avg_doc_length))  (:473-475), evaluated here ONCE
// per posting, operation by operation as Python evaluates it (tf is a Python int: it is converted, not truncated)
__global__ __launch_bounds__(256) void bm25_post_comp_kernel(const int32_t* __restrict__ post_doc,
                                                              const int32_t* __restrict__ post_tf,
                                                              const double* __restrict__ dnorm, double k1p1, int64_t n,
                                                              Bm25Post* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                   // the sentinel behind the last posting: "no posting here"
        Bm25Post z;
        z.doc = -1; z.comp_lo = 0u; z.comp_hi = 0u;
        out[n] = z;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int32_t d = post_doc[i], f = post_tf[i];
        const double tf = (double)f;
        const double comp = (tf * k1p1) / (tf + dnorm[d]);
        Bm25Post p;
        p.doc = d;
        p.comp_lo = (uint32_t)__double2loint(comp);
        p.comp_hi = (uint32_t)__double2hiint(comp);
        out[i] = p;
    }
}

// row h of the dense tables <- the tf_components of term dense_terms[h], by document (the rows are zero on entry: 0.0 marks
// a document without the term; a tf_component itself is never 0: tf >= 1, finite positive denominator)
__global__ __launch_bounds__(256) void bm25_dense_kernel(Bm25Index ix, const int32_t* __restrict__ dense_terms,
                                                          double* __restrict__ dense_comp, int64_t dense_stride) {
    const int h = blockIdx.y;
    const int32_t t = dense_terms[h];
    const int64_t s = ix.term_off[t], e = ix.term_off[t + 1];
    double* row = dense_comp + (int64_t)h * dense_stride;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = s + (int64_t)blockIdx.x * 256 + threadIdx.x; i < e; i += stride) {
        const Bm25Post p = ix.post[i];
        row[p.doc] = post_comp(p);
    }
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

hipError_t msr_bm25_post_comp(const int32_t* post_doc, const int32_t* post_tf, const double* dnorm, double k1, int64_t n,
                              Bm25Post* out, hipStream_t stream) {
    if (n < 0) return hipSuccess;
    bm25_post_comp_kernel<<<4096, 256, 0, stream>>>(post_doc, post_tf, dnorm, k1 + 1.0, n, out);   // self.k1 + 1
    return hipGetLastError();
}

hipError_t msr_bm25_build_dense(const Bm25Index& ix, const int32_t* dense_terms, int n_dense, double* dense_comp,
                                int64_t dense_stride, hipStream_t stream) {
    if (n_dense <= 0) return hipSuccess;
    bm25_dense_kernel<<<dim3(256, (unsigned)n_dense), 256, 0, stream>>>(ix, dense_terms, dense_comp, dense_stride);
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

int msr_bm25_max_segments(int64_t n_docs) { return (int)((n_docs + BM25_TILE - 1) / BM25_TILE); }

// An upper bound of every score of a query, as the 20-bit key prefix (sign, exponent, 8 mantissa bits of the order-preserving
// key) the select's window pass anchors its 4096 bins to: U = (k1 + 1) * sum over the query's terms with positive idf of
// idf * qtf (a tf_component is below k1 + 1), a little raised; out[q] = prefix(U) - 4094, i.e. U falls into bin 4094 and bin
// 4095 stays empty unless the bound is wrong -- which the select notices (it then takes its general path).
__global__ __launch_bounds__(256) void bm25_window_kernel(Bm25Index ix, const int32_t* __restrict__ q_term_off,
                                                           const int32_t* __restrict__ q_terms,
                                                           const int32_t* __restrict__ q_qtf, int q_first, int nq,
                                                           uint64_t* __restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int t0 = q_term_off[q_first + q];
    int nt = q_term_off[q_first + q + 1] - t0;
    if (nt > BM25_MAX_TERMS) nt = BM25_MAX_TERMS;
    double u = 0.0;
    for (int j = 0; j < nt; ++j) {
        const int32_t t = q_terms[t0 + j];
        if (t < 0 || t >= ix.n_terms) continue;
        const double w = (double)ix.idf[t] * (double)q_qtf[t0 + j];
        if (w > 0.0) u += w;
    }
    u = u * (ix.k1 + 1.0) * 1.0000001 + 1e-300;
    const uint64_t pre = msr_ord64(u) >> 44;
    out[q] = pre > 4094 ? pre - 4094 : 0;
}

hipError_t msr_bm25_window(const Bm25Index& ix, const int32_t* q_term_off, const int32_t* q_terms, const int32_t* q_qtf,
                           int q_first, int nq, uint64_t* out, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    bm25_window_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, stream>>>(ix, q_term_off, q_terms, q_qtf, q_first, nq, out);
    return hipGetLastError();
}

hipError_t msr_bm25_scores(const Bm25Index& ix, const int32_t* q_term_off, const int32_t* q_terms,
                           const int32_t* q_qtf, int q_first, int nq, double min_score, double* cand_score,
                           int32_t* cand_doc, int32_t* seg_n, int* n_seg, int64_t* seg_stride, hipStream_t stream) {
    *n_seg = 0; *seg_stride = 0;
    if (nq <= 0 || ix.n_docs <= 0) return hipSuccess;
    // A wave looks its query's terms up once and then walks `tpw` consecutive tiles with that plan in registers.  More tiles
    // per wave amortise the lookups (chains of dependent loads) but leave fewer work items: large batches take 8, a single
    // query one tile per wave (its ~1000 waves are all the parallelism it has).
    int tpw = BM25_TPW;
    while (tpw > 1 && (int64_t)nq * ((ix.n_tiles + tpw - 1) / tpw) < 8192) tpw >>= 1;
    const int n_spans = (ix.n_tiles + tpw - 1) / tpw;
    const int64_t items = (int64_t)nq * n_spans;
    if (items >= (1ll << 31)) return hipErrorInvalidValue;
    *n_seg = n_spans;
    *seg_stride = (int64_t)tpw * BM25_TILE;
    bm25_taat_kernel<<<(unsigned)((items + BM25_WAVES - 1) / BM25_WAVES), BM25_THREADS, 0, stream>>>(
        ix, q_term_off, q_terms, q_qtf, q_first, nq, min_score, tpw, n_spans, cand_score, cand_doc, seg_n, g_bm25_dbg);
    return hipGetLastError();
}
