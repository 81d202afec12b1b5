// K6 -- rerank / fuse of the stage-1 candidates (gfx950).  Compiled with -ffp-contract=off.
//
// Replaces the arithmetic of the /rerank endpoint (reranker/reranker_api.py):
//   candidate fetch   URL dedup (MIN(id) per URL-without-query-string), first <=10 chunks   :27-63
//   cosine            sklearn cosine_similarity, float32                                     :273-287
//   min-max           over ALL chunk rows of the request, python floats (float64)            :289-296, 360-361
//   blend             new*(1-smoothing) + old*smoothing                                      :362
//   positional        boost/decay of each document's best chunk                              :299-334
//   pool + order      per-document first maximum, descending                                 :370-372
// Two kernels: (A) one wave per (query, candidate) gathers the candidate's chunk rows and computes the
// cosines (HBM-bound gather of <= 10 x 3 KiB rows) plus three integers per candidate (rows, URL group,
// first row); (B) one workgroup per query runs the float64 chain on <= 1024 candidates entirely in LDS,
// reading ONLY the arrays (A) produced.  That split is what lets a doc-sharded index rerank a global
// candidate list: each shard runs (A) for the documents it owns, the halves of a query travel to the rank that fuses that
// query (one all-to-all, msretr/distributed.py) and are joined there (or_parts_kernel: all other shards contributed
// zeros), and that rank runs (B) for its share of the queries.
#include "msr_common.h"
#include "msr_internal.h"
#include "msr_sort.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RR_MAXC = 10;           // scratch slots per candidate (MSR_RERANK_MAX_CHUNKS)
constexpr int RR_THREADS = 1024;
constexpr int RR_MAXM = 1024;

// One wave per query and block of 8 candidate slots: lanes 0 .. 7 sort the slots into "mine" (the document lies in this
// shard) and "not mine" (zeros: the owner's words arrive through the join of the shards' halves), then the wave takes the
// slots that are mine one after the other.  (One wave per slot, as before, launches N times the waves a rank of an N-way
// sharded run has work for -- 2 M waves per 2048-query step at N = 8, most of which only wrote zeros: 2.2 ms instead of 0.7;
// workgroups of several waves sharing a block lose 12 % on the unsharded corpus, where every slot is work: a workgroup
// keeps its place on the CU until its slowest wave is done.)
// SLOTS = 1 (calls of a few queries, no records): one wave per candidate slot -- a single query has 1000 slots, i.e. 125 waves
// of 8 slots each on a chip with room for thousands; its gather took 43 us of one after the other.
constexpr int RC_SLOTS = 8;
template <bool TILED, int SLOTS = RC_SLOTS>
__global__ __launch_bounds__(64) void rerank_cos_kernel(DenseIndex ix, const int32_t* __restrict__ url_group,
                                                         const float* __restrict__ qn,
                                                         const int32_t* __restrict__ cand_doc,
                                                         const int32_t* __restrict__ cand_n, int max_cand,
                                                         int doc_base, int row_base, int max_chunks,
                                                         float* __restrict__ cos_out, int32_t* __restrict__ meta,
                                                         int q_per_block, int64_t block_stride, RerankRecords rec) {
    // cand_doc holds GLOBAL document indices; this shard owns [doc_base, doc_base + n_docs).
    // Output layout: query q's rows start (q / q_per_block) * block_stride words + (q % q_per_block) rows into cos_out / meta
    // (one contiguous array: q_per_block >= the number of queries; the blocks of an all-to-all send buffer: see msretr.h).
    // RECORDS (rec.out != null): nothing is written for the slots this shard does not own; an owned slot becomes a 16-word
    // record [slot, rows, url group + 2, first row, cos x 10, query, 0] at the place rerank_plan_kernel counted out for it.
    const int q = blockIdx.y, m0 = blockIdx.x * SLOTS, lane = threadIdx.x;
    const bool records = rec.out != nullptr;
    cos_out += (int64_t)(q / q_per_block) * block_stride + (int64_t)(q % q_per_block) * max_cand * RR_MAXC;
    meta += (int64_t)(q / q_per_block) * block_stride + (int64_t)(q % q_per_block) * max_cand * 3;
    int d_mine = -1;
    {
        const int m = m0 + lane;
        const bool slot = lane < SLOTS && m < max_cand;
        if (slot && m < cand_n[q]) d_mine = cand_doc[(int64_t)q * max_cand + m] - doc_base;
        const bool own = d_mine >= 0 && d_mine < ix.n_docs;
        if (!own) d_mine = -1;
        if (slot && !own && !records) {                  // not a candidate, or owned by another shard
            float* out = cos_out + (int64_t)m * RR_MAXC;
            int32_t* mt = meta + (int64_t)m * 3;
#pragma unroll
            for (int i = 0; i < RR_MAXC; ++i) out[i] = 0.f;
            mt[0] = 0; mt[1] = 0; mt[2] = 0;
        }
    }
    unsigned long long todo = __ballot(d_mine >= 0);
    if (todo == 0) return;
    const unsigned long long mine_all = todo;
    int64_t rec_first = 0;
    if (records) rec_first = (int64_t)rec.q_base[q] + rec.blk_off[(int64_t)q * gridDim.x + blockIdx.x];
    const f32x4* q4 = (const f32x4*)(qn + (size_t)q * MSR_DIM);
    const f32x4 qa = q4[lane], qb = q4[lane + 64], qc = q4[lane + 128];
    for (; todo != 0; todo &= todo - 1) {
        const int sl = __builtin_ctzll(todo);            // (wave-uniform)
        const int d = __builtin_amdgcn_readlane(d_mine, sl), m = m0 + sl;
        float* out = cos_out + (int64_t)m * RR_MAXC;
        int32_t* mt = meta + (int64_t)m * 3;
        if (records) {
            const int64_t at = rec_first + __builtin_popcountll(mine_all & ((1ull << sl) - 1));
            if (at < 0 || at >= rec.capacity) continue;      // (wave-uniform)
            int32_t* r = rec.out + at * MSR_RERANK_RECORD_WORDS;
            out = (float*)(r + 4);
            mt = r + 1;
            if (lane == 0) { r[0] = m; r[14] = q; r[15] = 0; }
        }
        const int64_t ds = ix.doc_off[d];
        int64_t de = ix.doc_off[d + 1];
        if (ds + max_chunks < de) de = ds + max_chunks;
        if (lane == 0) {
            mt[0] = (int32_t)(de - ds);                      // chunk rows that take part (<= max_chunks)
            // URL group + 2, so that after the join of the shards' halves 0 = nobody owns it, 1 = owned, not in urlsDB
            mt[1] = (url_group ? url_group[d] : d + doc_base) + 2;
            mt[2] = (int32_t)ds + row_base;                  // global row of the document's first chunk
        }
        if (lane >= (int)(de - ds) && lane < RR_MAXC) out[lane] = 0.f;
        // two rows in flight per wave: the loads of row c + 1 are issued before row c is reduced (a document's rows are
        // consecutive: 6 KB per pair)
        auto load_row = [&](int64_t c, f32x4& a, f32x4& b, f32x4& e2) {
            if (TILED) {
                const f32x4* base = (const f32x4*)(ix.emb + (size_t)(c >> 4) * (16 * MSR_DIM));
                const int i = (int)(c & 15);
                // float4 number v of the row (dims 4v..4v+3) lives at block t = v >> 2, lane 16 (v & 3) + i
                const int v0 = lane, v1 = lane + 64, v2 = lane + 128;
                a = base[(v0 >> 2) * 64 + (v0 & 3) * 16 + i];
                b = base[(v1 >> 2) * 64 + (v1 & 3) * 16 + i];
                e2 = base[(v2 >> 2) * 64 + (v2 & 3) * 16 + i];
            } else {
                const f32x4* p = (const f32x4*)(ix.emb + (size_t)c * MSR_DIM);
                a = p[lane]; b = p[lane + 64]; e2 = p[lane + 128];
            }
        };
        auto dot_row = [&](const f32x4& a, const f32x4& b, const f32x4& e2) {
            float s = a.x * qa.x + a.y * qa.y + a.z * qa.z + a.w * qa.w;
            s += b.x * qb.x + b.y * qb.y + b.z * qb.z + b.w * qb.w;
            s += e2.x * qc.x + e2.y * qc.y + e2.z * qc.z + e2.w * qc.w;
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            return s;
        };
        f32x4 a0, b0, e0, a1, b1, e1;
        if (ds < de) load_row(ds, a0, b0, e0);
        for (int64_t c = ds; c < de; c += 2) {
            const bool two = c + 1 < de;                     // (wave-uniform)
            if (two) load_row(c + 1, a1, b1, e1);
            const float s0 = dot_row(a0, b0, e0);
            if (c + 2 < de) load_row(c + 2, a0, b0, e0);
            if (lane == 0) out[c - ds] = s0 * ix.inv_norm[c];
            if (two) {
                const float s1 = dot_row(a1, b1, e1);
                if (lane == 0) out[c + 1 - ds] = s1 * ix.inv_norm[c + 1];
            }
        }
    }
}

__device__ __forceinline__ double block_reduce(double v, bool is_min, double* red) {
    // red: LDS scratch of RR_THREADS/64 doubles
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_xor(v, o);
        v = is_min ? (u < v ? u : v) : (u > v ? u : v);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < RR_THREADS / 64; ++i) {
        const double u = red[i];
        r = is_min ? (u < r ? u : r) : (u > r ? u : r);
    }
    return r;
}

__global__ __launch_bounds__(RR_THREADS) void rerank_fuse_kernel(
    const int32_t* __restrict__ cand_doc, const double* __restrict__ cand_bm25,
    const int32_t* __restrict__ cand_n, int max_cand, RerankParams prm, const float* __restrict__ cos_in,
    const int32_t* __restrict__ meta, int32_t* __restrict__ out_doc, double* __restrict__ out_score,
    double* __restrict__ out_orig, int32_t* __restrict__ out_chunk, int32_t* __restrict__ out_n,
    int32_t* __restrict__ out_rows) {
    __shared__ uint64_t khi[RR_MAXM];
    __shared__ uint32_t klo[RR_MAXM];
    __shared__ uint32_t val[RR_MAXM];
    __shared__ double sc[RR_MAXM];
    __shared__ double og[RR_MAXM];
    __shared__ int32_t ch[RR_MAXM];
    __shared__ uint8_t keep[RR_MAXM];
    __shared__ double red[RR_THREADS / 64];
    __shared__ int cnt[2];

    const int q = blockIdx.x, tid = threadIdx.x;
    int n = cand_n[q];
    if (n > max_cand) n = max_cand;
    if (n > RR_MAXM) n = RR_MAXM;
    int P = 64;
    while (P < n) P <<= 1;
    const int32_t* cd = cand_doc + (int64_t)q * max_cand;
    const double* cb = cand_bm25 + (int64_t)q * max_cand;
    const int32_t* mt = meta + (int64_t)q * max_cand * 3;      // (rows, url group, first row) per candidate

    // 1. sort candidates by (url group, doc) so the first entry of each group is MIN(id)   (:38-47)
    for (int i = tid; i < P; i += RR_THREADS) {
        uint64_t key = ~0ull;
        if (i < n) {
            const int d = cd[i];
            const int g = mt[3 * i + 1] - 2;
            if (d >= 0 && g >= 0) key = ((uint64_t)(uint32_t)g << 32) | (uint32_t)d;
        }
        khi[i] = key; klo[i] = 0; val[i] = (uint32_t)i;
    }
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    msr_sort::bitonic_sort<RR_THREADS, true>(khi, klo, val, P, true);

    // 2. keep = first of its URL group, has at least one chunk row
    double cmin = __builtin_inf(), cmax = -__builtin_inf(), bmin = __builtin_inf(), bmax = -__builtin_inf();
    int my_rows = 0;
    for (int i = tid; i < P; i += RR_THREADS) {                  // at most one iteration
        bool k = false;
        const uint64_t key = khi[i];
        if (key != ~0ull && (i == 0 || (khi[i - 1] >> 32) != (key >> 32))) {
            const int m = (int)val[i];
            const int nr = mt[3 * m];
            if (nr > 0) {
                k = true;
                const float* cs = cos_in + ((int64_t)q * max_cand + m) * RR_MAXC;
                for (int j = 0; j < (int)nr; ++j) {
                    const double c = (double)cs[j];
                    cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
                }
                const double bm = cb[m];
                bmin = bm < bmin ? bm : bmin; bmax = bm > bmax ? bm : bmax;
                my_rows += (int)nr;
            }
        }
        keep[i] = k ? 1 : 0;
    }
    cmin = block_reduce(cmin, true, red);
    cmax = block_reduce(cmax, false, red);
    bmin = block_reduce(bmin, true, red);
    bmax = block_reduce(bmax, false, red);
    if (my_rows) atomicAdd(&cnt[0], my_rows);
    __syncthreads();

    // 3. per kept document: normalise, blend, positional weighting, first maximum
    const double one_m_s = 1.0 - prm.smoothing;
    if (tid < P) {                                               // P <= RR_THREADS: one entry per thread
        const int i = tid;
        uint64_t shi = 0; uint32_t slo = 0;
        if (keep[i]) {
            const int d = (int)(uint32_t)khi[i];
            const int m = (int)val[i];
            const int nr = mt[3 * m];
            const float* cs = cos_in + ((int64_t)q * max_cand + m) * RR_MAXC;
            const double old = (bmax == bmin) ? 0.0 : (cb[m] - bmin) / (bmax - bmin);
            double v[RR_MAXC];
            int best = 0;
            for (int j = 0; j < nr; ++j) {
                const double nw = (cmax == cmin) ? 0.0 : ((double)cs[j] - cmin) / (cmax - cmin);
                v[j] = nw * one_m_s + old * prm.smoothing;
                if (v[j] > v[best]) best = j;
            }
            if (nr > 1) {
                const double ratio = (double)best / (double)(nr - 1);
                const double adj = prm.max_boost - (prm.max_boost + prm.max_decay) * ratio;
                double a = v[best] + adj;
                a = a < 1.0 ? a : 1.0;                           // min(1.0, adjusted)
                a = a > 0.0 ? a : 0.0;                           // max(0.0, ...)
                v[best] = a;
                best = 0;
                for (int j = 1; j < nr; ++j)
                    if (v[j] > v[best]) best = j;
            }
            sc[i] = v[best]; og[i] = old; ch[i] = mt[3 * m + 2] + best;
            shi = msr_ord64(v[best]); slo = ~(uint32_t)d;
            atomicAdd(&cnt[1], 1);
        }
        khi[i] = shi; klo[i] = slo; val[i] = (uint32_t)i;
    }
    __syncthreads();
    // 4. order by (score desc, doc asc)
    msr_sort::bitonic_sort<RR_THREADS, true>(khi, klo, val, P, false);
    const int n_keep = cnt[1];
    for (int i = tid; i < max_cand; i += RR_THREADS) {
        const int64_t o = (int64_t)q * max_cand + i;
        if (i < n_keep) {
            const int src = (int)val[i];
            out_doc[o] = (int32_t)~klo[i]; out_score[o] = sc[src]; out_orig[o] = og[src]; out_chunk[o] = ch[src];
        } else {
            out_doc[o] = -1; out_score[o] = -__builtin_inf(); out_orig[o] = 0.0; out_chunk[o] = -1;
        }
    }
    if (tid == 0) { out_n[q] = n_keep; out_rows[q] = cnt[0]; }
}


// ---- domain diversification of the fused lists (reranker_api.py:178-236), one workgroup per query ----------------------
// Input: the fused list of a query as rerank_fuse_kernel leaves it -- (new_similarity desc, document asc) -- and one word
// per document: its DOMAIN id (urlparse(url).netloc.lower(), :170-176, numbered by the host at bind) or -1 for a document
// the reference's response models reject (NULL title / url / text, :376-397; dropped BEFORE the diversification).
// hybrid_diversification on a list in that order reduces to (derivation in DESIGN.md section 3, K6b):
//   * a domain is "high" iff its FIRST entry scores >= threshold (the list is sorted, the first entry is the domain's best);
//     every entry of a domain is in the same tier, so apply_domain_cap(.., 1) on either tier keeps exactly the first entry
//     of each domain: kept-high entries (score >= threshold) precede kept-medium ones in list order;
//   * final = kept-high + kept-medium[: top_k - #kept-high]   (Python slice semantics, a NEGATIVE bound included);
//   * if that is short of top_k: the dropped entries, stably sorted by score with the high tier's drops first on ties, fill
//     it up with their scores shifted by delta = first_dropped - last_kept + 1e-4, clamped at 0 -- all float64, the
//     reference's operations in the reference's order (the file is compiled with -ffp-contract=off);
//   * the closing sorted() is then the identity (kept scores descend, shifted scores descend from last_kept - 1e-4).
// diversify == 0 (config.yaml similarity.diversification false): the first top_k accepted entries.
constexpr int DV_THREADS = 1024;

__device__ __forceinline__ int dv_excl_scan(bool flag, int* wsum, int* total) {
    // exclusive prefix count of `flag` over the workgroup (thread order); *total = the count.  Two barriers.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                                     // (wsum may still be read from an earlier scan)
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < DV_THREADS / 64; ++i) {
        const int c = wsum[i];
        if (i < wv) base += c;
        tot += c;
    }
    *total = tot;
    return base + within;
}

__global__ __launch_bounds__(DV_THREADS) void diversify_kernel(
    const int32_t* __restrict__ f_doc, const double* __restrict__ f_score, const double* __restrict__ f_orig,
    const int32_t* __restrict__ f_chunk, const int32_t* __restrict__ f_n, int max_cand,
    const int32_t* __restrict__ doc_domain, int64_t n_domain_docs, int top_k, double threshold, int diversify,
    int32_t* __restrict__ out_doc, double* __restrict__ out_score, double* __restrict__ out_orig,
    int32_t* __restrict__ out_chunk, int32_t* __restrict__ out_n) {
    __shared__ int32_t Ldom[RR_MAXM];
    __shared__ double Ls[RR_MAXM];
    __shared__ int16_t Lsrc[RR_MAXM];
    __shared__ uint8_t Lhi[RR_MAXM], Ldrop[RR_MAXM];
    __shared__ int wsum[DV_THREADS / 64];
    __shared__ double s_last, s_a0;
    const int q = blockIdx.x, tid = threadIdx.x;
    int n = f_n[q];
    n = n < 0 ? 0 : (n > max_cand ? max_cand : n);
    n = n > RR_MAXM ? RR_MAXM : n;
    const int64_t row = (int64_t)q * max_cand;
    // 1. accepted entries, in list order
    int dm = -1;
    double sc = 0.0;
    if (tid < n) {
        const int d = f_doc[row + tid];
        sc = f_score[row + tid];
        if (d >= 0) dm = doc_domain ? ((int64_t)d < n_domain_docs ? doc_domain[d] : -1) : d;     // (no table: every document its own domain)
    }
    int nv = 0;
    const int v = dv_excl_scan(dm >= 0, wsum, &nv);
    if (dm >= 0) { Ldom[v] = dm; Ls[v] = sc; Lsrc[v] = (int16_t)tid; }
    __syncthreads();
    auto emit = [&](int pos, int src, double score) {
        out_doc[row + pos] = f_doc[row + src]; out_score[row + pos] = score;
        out_orig[row + pos] = f_orig[row + src]; out_chunk[row + pos] = f_chunk[row + src];
    };
    int n_out = 0;
    if (!diversify) {
        n_out = nv < top_k ? nv : top_k;
        if (tid < n_out) emit(tid, Lsrc[tid], Ls[tid]);
    } else {
        // 2. first entry of each domain; tier of the domain
        bool kept = false, hi = false;
        if (tid < nv) {
            const int mine = Ldom[tid];
            int f = 0;
            while (Ldom[f] != mine) ++f;                 // (terminates at f == tid at the latest)
            kept = f == tid;
            hi = Ls[f] >= threshold;
            Lhi[tid] = hi ? 1 : 0; Ldrop[tid] = kept ? 0 : 1;
        }
        int n_kept = 0, n_hi = 0;
        const int kr = dv_excl_scan(kept, wsum, &n_kept);
        (void)dv_excl_scan(kept && hi, wsum, &n_hi);
        const int n_med = n_kept - n_hi, remaining = top_k - n_hi;
        const int take_med = remaining >= 0 ? (n_med < remaining ? n_med : remaining) : (n_med + remaining > 0 ? n_med + remaining : 0);
        const int n_final = n_hi + take_med;
        const bool in_final = kept && kr < n_final;       // (kept-high entries have kr < n_hi, kept-medium ones kr - n_hi < take_med)
        if (in_final) {
            emit(kr, Lsrc[tid], Ls[tid]);
            if (kr == n_final - 1) s_last = Ls[tid];
        }
        n_out = n_final;
        const int need = top_k - n_final, n_drop = nv - n_kept;
        if (need > 0 && n_drop > 0) {                     // (workgroup-uniform)
            // 3. the dropped entries in the order of sorted(dropped_high + dropped_medium, key=score, reverse=True)
            int rr = 0;
            const bool dropped = tid < nv && !kept;
            if (dropped) {
                const double ms = Ls[tid];
                for (int j = 0; j < nv; ++j) {
                    if (!Ldrop[j] || j == tid) continue;
                    const double js = Ls[j];
                    const bool before = js > ms || (js == ms && (Lhi[j] > (uint8_t)hi || (Lhi[j] == (uint8_t)hi && j < tid)));
                    rr += before ? 1 : 0;
                }
                if (rr == 0) s_a0 = Ls[tid];
            }
            __syncthreads();
            const int n_add = need < n_drop ? need : n_drop;
            if (dropped && rr < n_add) {
                const double delta = (s_a0 - s_last) + 1e-4;
                const double x = Ls[tid] - delta;
                emit(n_final + rr, Lsrc[tid], x > 0.0 ? x : 0.0);
            }
            n_out = n_final + n_add;
        }
    }
    for (int i = n_out + tid; i < max_cand; i += DV_THREADS) {
        out_doc[row + i] = -1; out_score[row + i] = -__builtin_inf(); out_orig[row + i] = 0.0; out_chunk[row + i] = -1;
    }
    if (tid == 0) out_n[q] = n_out;
}

// ---- the sharded rerank's compact exchange ------------------------------------------------------------------------------
// A rank of an N-way run owns ~1/N of a query's candidates: the dense halves of msr_rerank_gather_blocks are 13 words per
// SLOT, mostly zeros.  Compact: a 16-word record per OWNED slot, variable counts -- which every rank can work out for every
// (source, query) without talking to anyone, because the merged candidate lists are replicated and the shards are document
// ranges.  rerank_plan_kernel counts, rerank_offsets_kernel turns the counts into the places of the records in the send and
// receive buffers and into the N x N matrix of records per (source, destination) the host sizes the all-to-all with.
constexpr int RP_THREADS = 256;
__global__ __launch_bounds__(RP_THREADS) void rerank_plan_kernel(const int32_t* __restrict__ cand_doc,
                                                                  const int32_t* __restrict__ cand_n, int max_cand,
                                                                  const int32_t* __restrict__ bounds, int n_shards, int my,
                                                                  int nq, int32_t* __restrict__ counts,
                                                                  int32_t* __restrict__ blk_off) {
    __shared__ int s_cnt[64];
    __shared__ int s_blk[RR_MAXM / RC_SLOTS + 1];
    const int q = blockIdx.x, t = threadIdx.x;
    const int n_blk = (max_cand + RC_SLOTS - 1) / RC_SLOTS;
    if (t < 64) s_cnt[t] = 0;
    __syncthreads();
    int n = cand_n[q];
    if (n > max_cand) n = max_cand;
    const int lo_my = bounds[my], hi_my = bounds[my + 1];
    for (int m0 = 0; m0 < n_blk * RC_SLOTS; m0 += RP_THREADS) {           // (whole waves: the ballot below)
        const int m = m0 + t;
        int owner = -1;
        const int d = m < n ? cand_doc[(int64_t)q * max_cand + m] : -1;
        if (d >= bounds[0] && d < bounds[n_shards]) {
            int a = 0, b = n_shards;                                       // bounds[a] <= d < bounds[b]
            while (b - a > 1) {
                const int c = (a + b) >> 1;
                if (d >= bounds[c]) a = c; else b = c;
            }
            owner = a;
        }
        if (owner >= 0) atomicAdd(&s_cnt[owner], 1);
        const unsigned long long mine = __ballot(d >= lo_my && d < hi_my);
        if ((t & (RC_SLOTS - 1)) == 0 && m < n_blk * RC_SLOTS)
            s_blk[m / RC_SLOTS] = __builtin_popcountll((mine >> (t & 63)) & ((1ull << RC_SLOTS) - 1));
    }
    __syncthreads();
    if (t < n_shards) counts[(int64_t)t * nq + q] = s_cnt[t];
    if (t == 0) {                                                          // <= 128 blocks: a serial prefix in LDS
        int acc = 0;
        for (int b = 0; b < n_blk; ++b) { const int c = s_blk[b]; s_blk[b] = acc; acc += c; }
    }
    __syncthreads();
    for (int b = t; b < n_blk; b += RP_THREADS) blk_off[(int64_t)q * n_blk + b] = s_blk[b];
}

constexpr int RO_THREADS = 1024;
// exclusive prefix of in[0 .. n) + base -> out, by one workgroup
__device__ void block_excl_scan(const int32_t* __restrict__ in, int n, int base, int32_t* __restrict__ out, int* sh) {
    const int t = threadIdx.x, per = (n + RO_THREADS - 1) / RO_THREADS;
    const int a = t * per < n ? t * per : n, b = a + per < n ? a + per : n;
    int sum = 0;
    for (int i = a; i < b; ++i) sum += in[i];
    sh[t] = sum;
    __syncthreads();
    for (int o = 1; o < RO_THREADS; o <<= 1) {
        const int v = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    int acc = base + sh[t] - sum;
    for (int i = a; i < b; ++i) { out[i] = acc; acc += in[i]; }
    __syncthreads();
}
// sum of in[0 .. n) by ONE wave (every lane gets it)
__device__ __forceinline__ int wave_sum(const int32_t* __restrict__ in, int n) {
    int sum = 0;
    for (int i = threadIdx.x & 63; i < n; i += 64) sum += in[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    return sum;
}
// block s < n_shards: the places of source s' records for MY queries in the receive buffer; block n_shards: the places of my
// records in the send buffer; block n_shards + 1: pair[s][o] = records source s has for the queries of rank o (a wave per
// pair, no barrier).
__global__ __launch_bounds__(RO_THREADS) void rerank_offsets_kernel(const int32_t* __restrict__ counts, int n_shards, int my,
                                                                     int nq, int qps, int32_t* __restrict__ send_base,
                                                                     int32_t* __restrict__ recv_off,
                                                                     int32_t* __restrict__ pair) {
    __shared__ int sh[RO_THREADS];
    __shared__ int s_base[64];
    const int b = blockIdx.x, w = threadIdx.x >> 6;
    const int lo = my * qps < nq ? my * qps : nq, hi = lo + qps < nq ? lo + qps : nq;
    if (b < n_shards) {
        for (int s = w; s < b; s += RO_THREADS / 64) {                     // what the sources before b send me
            const int v = wave_sum(counts + (int64_t)s * nq + lo, hi - lo);
            if ((threadIdx.x & 63) == 0) s_base[s] = v;
        }
        __syncthreads();
        int base = 0;
        for (int s = 0; s < b; ++s) base += s_base[s];
        block_excl_scan(counts + (int64_t)b * nq + lo, hi - lo, base, recv_off + (int64_t)b * qps, sh);
    } else if (b == n_shards) {
        block_excl_scan(counts + (int64_t)my * nq, nq, 0, send_base, sh);
    } else {
        for (int p = w; p < n_shards * n_shards; p += RO_THREADS / 64) {
            const int s = p / n_shards, o = p - s * n_shards;
            const int a0 = o * qps < nq ? o * qps : nq, a1 = a0 + qps < nq ? a0 + qps : nq;
            const int v = wave_sum(counts + (int64_t)s * nq + a0, a1 - a0);
            if ((threadIdx.x & 63) == 0) pair[p] = v;
        }
    }
}

// The receiving side: the records of (source s, my query j) go to their slots of the dense arrays the fuse kernel reads
// (zeroed beforehand: a slot nobody owns stays "no document").  One workgroup per (query, source).
__global__ __launch_bounds__(256) void rerank_scatter_kernel(const int32_t* __restrict__ records, int64_t capacity,
                                                              const int32_t* __restrict__ counts,
                                                              const int32_t* __restrict__ recv_off, int nq, int qps, int q_first,
                                                              int max_cand, float* __restrict__ cos_out,
                                                              int32_t* __restrict__ meta_out) {
    const int j = blockIdx.x, s = blockIdx.y;
    int cnt = counts[(int64_t)s * nq + q_first + j];
    const int64_t first = recv_off[(int64_t)s * qps + j];
    if (first < 0 || first >= capacity) return;
    if (first + cnt > capacity) cnt = (int)(capacity - first);
    const int32_t* rec = records + first * MSR_RERANK_RECORD_WORDS;
    for (int i = threadIdx.x; i < cnt * 13; i += 256) {
        const int r = i / 13, w = i - 13 * r;
        const int32_t* rr = rec + (int64_t)r * MSR_RERANK_RECORD_WORDS;
        const int slot = rr[0];
        if (slot < 0 || slot >= max_cand) continue;                       // (never, with records this library wrote)
        if (w < 3) meta_out[((int64_t)j * max_cand + slot) * 3 + w] = rr[1 + w];
        else cos_out[((int64_t)j * max_cand + slot) * RR_MAXC + (w - 3)] = __int_as_float(rr[4 + (w - 3)]);
    }
}

}  // namespace

hipError_t msr_rerank_plan_run(int nq, const int32_t* cand_doc, const int32_t* cand_n, int max_cand, const int32_t* bounds,
                               int n_shards, int my, int qps, int32_t* counts, int32_t* send_base, int32_t* blk_off,
                               int32_t* recv_off, int32_t* pair, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if (max_cand <= 0 || max_cand > RR_MAXM || n_shards < 1 || n_shards > 64 || my < 0 || my >= n_shards || qps < 1)
        return hipErrorInvalidValue;
    rerank_plan_kernel<<<nq, RP_THREADS, 0, stream>>>(cand_doc, cand_n, max_cand, bounds, n_shards, my, nq, counts, blk_off);
    rerank_offsets_kernel<<<n_shards + 2, RO_THREADS, 0, stream>>>(counts, n_shards, my, nq, qps, send_base, recv_off, pair);
    return hipGetLastError();
}

hipError_t msr_rerank_scatter_run(const int32_t* records, int64_t capacity, const int32_t* counts, const int32_t* recv_off, int n_shards, int nq,
                                  int qps, int q_first, int n_mine, int max_cand, float* cos_out, int32_t* meta_out,
                                  hipStream_t stream) {
    if (n_mine <= 0) return hipSuccess;
    if (max_cand <= 0 || max_cand > RR_MAXM || n_shards < 1 || n_shards > 64) return hipErrorInvalidValue;
    hipError_t err = hipMemsetAsync(cos_out, 0, (size_t)n_mine * max_cand * RR_MAXC * 4, stream);
    if (err != hipSuccess) return err;
    err = hipMemsetAsync(meta_out, 0, (size_t)n_mine * max_cand * 3 * 4, stream);
    if (err != hipSuccess) return err;
    rerank_scatter_kernel<<<dim3((unsigned)n_mine, (unsigned)n_shards), 256, 0, stream>>>(records, capacity, counts, recv_off, nq, qps,
                                                                                           q_first, max_cand, cos_out, meta_out);
    return hipGetLastError();
}

hipError_t msr_diversify_run(int nq, const int32_t* f_doc, const double* f_score, const double* f_orig, const int32_t* f_chunk,
                             const int32_t* f_n, int max_cand, const int32_t* doc_domain, int64_t n_domain_docs, int top_k,
                             double threshold, int diversify, int32_t* out_doc, double* out_score, double* out_orig,
                             int32_t* out_chunk, int32_t* out_n, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if (max_cand <= 0 || max_cand > RR_MAXM || top_k < 1) return hipErrorInvalidValue;
    diversify_kernel<<<nq, DV_THREADS, 0, stream>>>(f_doc, f_score, f_orig, f_chunk, f_n, max_cand, doc_domain, n_domain_docs,
                                                    top_k, threshold, diversify, out_doc, out_score, out_orig, out_chunk, out_n);
    return hipGetLastError();
}

hipError_t msr_rerank_gather(const DenseIndex& ix, const int32_t* url_group, const float* qn, int nq,
                             const int32_t* cand_doc, const int32_t* cand_n, int max_cand, int doc_base,
                             int row_base, int max_chunks, float* cos_out, int32_t* meta, int q_per_block,
                             int64_t block_stride, const RerankRecords& rec, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if (max_cand <= 0 || max_cand > RR_MAXM || max_chunks <= 0 || max_chunks > RR_MAXC || q_per_block < 1) return hipErrorInvalidValue;
    if (rec.out == nullptr && (int64_t)nq * max_cand <= 16384) {      // a few queries: one wave per slot
        dim3 grid1((unsigned)max_cand, (unsigned)nq);
        if (ix.layout == 1)
            rerank_cos_kernel<true, 1><<<grid1, 64, 0, stream>>>(ix, url_group, qn, cand_doc, cand_n, max_cand, doc_base,
                                                                 row_base, max_chunks, cos_out, meta, q_per_block, block_stride, rec);
        else
            rerank_cos_kernel<false, 1><<<grid1, 64, 0, stream>>>(ix, url_group, qn, cand_doc, cand_n, max_cand, doc_base,
                                                                  row_base, max_chunks, cos_out, meta, q_per_block, block_stride, rec);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((max_cand + RC_SLOTS - 1) / RC_SLOTS), (unsigned)nq);
    if (ix.layout == 1)
        rerank_cos_kernel<true><<<grid, 64, 0, stream>>>(ix, url_group, qn, cand_doc, cand_n, max_cand, doc_base,
                                                         row_base, max_chunks, cos_out, meta, q_per_block, block_stride, rec);
    else
        rerank_cos_kernel<false><<<grid, 64, 0, stream>>>(ix, url_group, qn, cand_doc, cand_n, max_cand, doc_base,
                                                          row_base, max_chunks, cos_out, meta, q_per_block, block_stride, rec);
    return hipGetLastError();
}

// out[i] = OR over parts p of in[p][i]: the join of the per-shard halves of msr_rerank_gather.  Exactly one shard wrote a
// non-zero word for a candidate (the owner of its document), every other shard wrote 0, so the OR -- like the integer sum the
// all-reduce form used -- IS that shard's word.  16 bytes per thread and step.
__global__ __launch_bounds__(256) void or_parts_kernel(const uint4* __restrict__ in, int n_parts, int64_t part_stride16,
                                                        int64_t n16, uint4* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        uint4 v = in[i];
        for (int p = 1; p < n_parts; ++p) {
            const uint4 u = in[p * part_stride16 + i];
            v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
        }
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void or_parts_tail_kernel(const uint32_t* __restrict__ in, int n_parts, int64_t part_stride,
                                                             int64_t first, int64_t n, uint32_t* __restrict__ out) {
    const int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t v = in[i];
    for (int p = 1; p < n_parts; ++p) v |= in[p * part_stride + i];
    out[i] = v;
}

hipError_t msr_rerank_fuse_run(int nq, const int32_t* cand_doc, const double* cand_bm25, const int32_t* cand_n,
                               int max_cand, const RerankParams& p, const float* cos_in, const int32_t* meta,
                               int32_t* out_doc, double* out_score, double* out_orig, int32_t* out_chunk,
                               int32_t* out_n, int32_t* out_rows, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if (max_cand <= 0 || max_cand > RR_MAXM || p.max_chunks <= 0 || p.max_chunks > RR_MAXC) return hipErrorInvalidValue;
    rerank_fuse_kernel<<<nq, RR_THREADS, 0, stream>>>(cand_doc, cand_bm25, cand_n, max_cand, p, cos_in, meta, out_doc,
                                                      out_score, out_orig, out_chunk, out_n, out_rows);
    return hipGetLastError();
}

// out[0 .. n_words) = OR over the n_parts arrays in + p * part_stride_bytes (32-bit words; every pointer and the stride
// 4-byte aligned, 16-byte aligned ones take the wide path)
hipError_t msr_or_parts(const void* in, int n_parts, int64_t part_stride_bytes, int64_t n_words, void* out, hipStream_t stream) {
    if (n_words <= 0) return hipSuccess;
    const bool wide = (((uintptr_t)in | (uintptr_t)out | (uint64_t)part_stride_bytes) & 15) == 0;
    const int64_t n16 = wide ? n_words / 4 : 0;
    if (n16 > 0) {
        const unsigned grid = (unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
        or_parts_kernel<<<grid, 256, 0, stream>>>((const uint4*)in, n_parts, part_stride_bytes / 16, n16, (uint4*)out);
    }
    const int64_t rest = n_words - 4 * n16;
    if (rest > 0)
        or_parts_tail_kernel<<<(unsigned)((rest + 255) / 256), 256, 0, stream>>>((const uint32_t*)in, n_parts, part_stride_bytes / 4,
                                                                              4 * n16, n_words, (uint32_t*)out);
    return hipGetLastError();
}
