/* CPU restatement (plain C) of the reference's BM25 scoring loop + sort + cut.
 * TEST INFRASTRUCTURE ONLY: used by tests and by bench.py's cpu_baseline leg, never by the product.
 * Follows /root/reference/indexer/bm25_indexer.py:458-488 (float64 accumulation in query-term order,
 * candidates = documents touched by a posting with score >= min_score, full sort, ties by ascending doc).
 * Compile with -ffp-contract=off: Python never fuses a*b+c. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double s; int32_t d; } cand_t;

static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->d > y->d) - (x->d < y->d);
}

/* returns the number of results written (<= k), or -1 on allocation failure */
int orc_bm25_topk(const int64_t* term_off, const int32_t* post_doc, const int32_t* post_tf,
                  const int32_t* doc_len, const float* idf, float avgdl_f32, double k1, double b,
                  int64_t n_docs, int64_t n_terms, const int32_t* q_terms, const int32_t* q_qtf, int n_q_terms,
                  int k, double min_score, int32_t* out_doc, double* out_score) {
    double* acc = (double*)calloc((size_t)n_docs, sizeof(double));
    uint8_t* touched = (uint8_t*)calloc((size_t)n_docs, 1);
    if (!acc || !touched) { free(acc); free(touched); return -1; }
    const double avgdl = (double)avgdl_f32;
    int64_t n_cand_max = 0;
    for (int j = 0; j < n_q_terms; ++j) {
        const int32_t t = q_terms[j];
        if (t < 0 || t >= n_terms) continue;
        const double idf_t = (double)idf[t];
        const double qtf = (double)q_qtf[j];
        for (int64_t i = term_off[t]; i < term_off[t + 1]; ++i) {
            const int32_t d = post_doc[i];
            const double tf = (double)post_tf[i];
            const double dl = (double)doc_len[d];
            const double comp = (tf * (k1 + 1.0)) / (tf + k1 * ((1.0 - b) + (b * dl) / avgdl));
            acc[d] = acc[d] + (idf_t * comp) * qtf;
            if (!touched[d]) { touched[d] = 1; ++n_cand_max; }
        }
    }
    cand_t* c = (cand_t*)malloc((size_t)(n_cand_max > 0 ? n_cand_max : 1) * sizeof(cand_t));
    if (!c) { free(acc); free(touched); return -1; }
    int64_t n = 0;
    for (int64_t d = 0; d < n_docs; ++d)
        if (touched[d] && acc[d] >= min_score) { c[n].s = acc[d]; c[n].d = (int32_t)d; ++n; }
    qsort(c, (size_t)n, sizeof(cand_t), cmp_cand);          /* the reference sorts ALL candidates (:484) */
    const int m = n < k ? (int)n : k;
    for (int i = 0; i < m; ++i) { out_doc[i] = c[i].d; out_score[i] = c[i].s; }
    free(c); free(acc); free(touched);
    return m;
}
