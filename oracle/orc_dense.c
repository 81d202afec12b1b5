/* CPU restatement (plain C + OpenMP) of the dense full scan: cosine of one query against every chunk,
 * per-document max, top-k.  TEST INFRASTRUCTURE ONLY (tests, bench.py cpu_baseline).
 * Cosine as sklearn.metrics.pairwise.cosine_similarity computes it in float32
 * (/root/reference/reranker/reranker_api.py:285): both sides divided by their L2 norm (0 -> 1), then dot.
 * Per-document max = reranker_api.py:370 / Project_Report p.2; order = score desc, doc index asc. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DIM 768

typedef struct { float s; int32_t d; } cand_t;

static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->d > y->d) - (x->d < y->d);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* best[n_docs] (-inf for chunk-less docs), arg[n_docs] (row of first maximum or -1) */
void orc_dense_doc_scores(const float* emb, const int32_t* doc_off, int64_t n_docs, const float* q,
                          int max_chunks, float* best, int32_t* arg) {
    float qn[DIM];
    float ss = 0.f;
    for (int j = 0; j < DIM; ++j) ss += q[j] * q[j];
    float nq = sqrtf(ss);
    if (nq == 0.f) nq = 1.f;
    for (int j = 0; j < DIM; ++j) qn[j] = q[j] / nq;
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n_docs; ++d) {
        int64_t lo = doc_off[d], hi = doc_off[d + 1];
        if (max_chunks > 0 && lo + max_chunks < hi) hi = lo + max_chunks;
        float bs = -INFINITY;
        int32_t ba = -1;
        for (int64_t c = lo; c < hi; ++c) {
            const float* e = emb + (size_t)c * DIM;
            float dot = 0.f, nn = 0.f;
            for (int j = 0; j < DIM; ++j) { dot += e[j] * qn[j]; nn += e[j] * e[j]; }
            float ne = sqrtf(nn);
            if (ne == 0.f) ne = 1.f;
            const float s = dot / ne;
            if (s > bs) { bs = s; ba = (int32_t)c; }
        }
        best[d] = bs;
        arg[d] = ba;
    }
}

int orc_dense_topk(const float* emb, const int32_t* doc_off, int64_t n_docs, const float* q, int k,
                   int max_chunks, int32_t* out_doc, float* out_score, int32_t* out_chunk) {
    float* best = (float*)malloc((size_t)n_docs * sizeof(float));
    int32_t* arg = (int32_t*)malloc((size_t)n_docs * sizeof(int32_t));
    cand_t* c = (cand_t*)malloc((size_t)n_docs * sizeof(cand_t));
    if (!best || !arg || !c) { free(best); free(arg); free(c); return -1; }
    orc_dense_doc_scores(emb, doc_off, n_docs, q, max_chunks, best, arg);
    int64_t n = 0;
    for (int64_t d = 0; d < n_docs; ++d)
        if (best[d] > -INFINITY) { c[n].s = best[d]; c[n].d = (int32_t)d; ++n; }
    qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);
    const int m = n < k ? (int)n : k;
    for (int i = 0; i < m; ++i) { out_doc[i] = c[i].d; out_score[i] = c[i].s; out_chunk[i] = arg[c[i].d]; }
    free(best); free(arg); free(c);
    return m;
}
