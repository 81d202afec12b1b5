/* CPU restatement (plain C) of the reference's BM25 scoring loop + sort + cut.
 * TEST INFRASTRUCTURE ONLY: used by tests and by bench.py's cpu_baseline leg, never by the product.
 * Follows /root/reference/indexer/bm25_indexer.py:458-488 (float64 accumulation in query-term order,
 * candidates = documents touched by a posting with score >= min_score, full sort, ties by ascending doc).
 * Compile with -ffp-contract=off: Python never fuses a*b+c. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double s; int32_t d; } cand_t;

static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->d > y->d) - (x->d < y->d);
}

static int64_t lower_bound_i32(const int32_t* a, int64_t lo, int64_t hi, int32_t target) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* returns the number of results written (<= k), or -1 on allocation failure.
 * OpenMP: the documents are cut into one range per thread; a thread walks, term by term IN QUERY ORDER, the slice of
 * every posting list that falls into its range, so each document's float64 sum is formed by one thread in the
 * reference's order (:466-478) and the result does not depend on the thread count. */
int orc_bm25_topk(const int64_t* term_off, const int32_t* post_doc, const int32_t* post_tf,
                  const int32_t* doc_len, const float* idf, float avgdl_f32, double k1, double b,
                  int64_t n_docs, int64_t n_terms, const int32_t* q_terms, const int32_t* q_qtf, int n_q_terms,
                  int k, double min_score, int32_t* out_doc, double* out_score) {
    double* acc = (double*)calloc((size_t)n_docs, sizeof(double));
    uint8_t* touched = (uint8_t*)calloc((size_t)n_docs, 1);
    if (!acc || !touched) { free(acc); free(touched); return -1; }
    const double avgdl = (double)avgdl_f32;
    int n_threads = 1;
#ifdef _OPENMP
    n_threads = omp_get_max_threads();
    if (n_docs < 65536) n_threads = 1;
#endif
    int64_t* counts = (int64_t*)calloc((size_t)n_threads + 1, sizeof(int64_t));
    if (!counts) { free(acc); free(touched); return -1; }
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        const int32_t d_lo = (int32_t)(n_docs * tid / n_threads), d_hi = (int32_t)(n_docs * (tid + 1) / n_threads);
        for (int j = 0; j < n_q_terms; ++j) {
            const int32_t t = q_terms[j];
            if (t < 0 || t >= n_terms) continue;
            const double idf_t = (double)idf[t];
            const double qtf = (double)q_qtf[j];
            const int64_t s = lower_bound_i32(post_doc, term_off[t], term_off[t + 1], d_lo);
            const int64_t e = lower_bound_i32(post_doc, s, term_off[t + 1], d_hi);
            for (int64_t i = s; i < e; ++i) {
                const int32_t d = post_doc[i];
                const double tf = (double)post_tf[i];
                const double dl = (double)doc_len[d];
                const double comp = (tf * (k1 + 1.0)) / (tf + k1 * ((1.0 - b) + (b * dl) / avgdl));
                acc[d] = acc[d] + (idf_t * comp) * qtf;
                touched[d] = 1;
            }
        }
        int64_t c = 0;
        for (int32_t d = d_lo; d < d_hi; ++d) c += touched[d] && acc[d] >= min_score;
        counts[tid + 1] = c;
    }
    for (int t = 0; t < n_threads; ++t) counts[t + 1] += counts[t];
    const int64_t n = counts[n_threads];
    cand_t* c = (cand_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(cand_t));
    if (!c) { free(acc); free(touched); free(counts); return -1; }
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        const int32_t d_lo = (int32_t)(n_docs * tid / n_threads), d_hi = (int32_t)(n_docs * (tid + 1) / n_threads);
        int64_t o = counts[tid];
        for (int32_t d = d_lo; d < d_hi; ++d)
            if (touched[d] && acc[d] >= min_score) { c[o].s = acc[d]; c[o].d = d; ++o; }
    }
    qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);          /* the reference sorts ALL candidates (:484) */
    const int m = n < k ? (int)n : k;
    for (int i = 0; i < m; ++i) { out_doc[i] = c[i].d; out_score[i] = c[i].s; }
    free(c); free(acc); free(touched); free(counts);
    return m;
}
