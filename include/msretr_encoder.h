/* libmsretr -- query encoder pieces (same library, same conventions as msretr.h).
 *
 * The reference turns the query string into the 768-float vector the retriever consumes with a sentence-transformers
 * bi-encoder, `embedding_model.encode(request.query)` (reranker/reranker_api.py:137-139,355; index side
 * indexer/indexer.py:165 with normalize_embeddings=True): a ModernBERT-base transformer + mean pooling.  The weights
 * are fetched by NAME there; here they come from a local directory (or are random for tests), see
 * modern-search-engines-project_amd/encoder.py.
 *
 * For a query (<= 128 tokens) the whole forward pass is hand-written HIP behind the entry points below: the four matrix
 * products of a layer (msr_enc_linear, exact-f32 matrix cores, weights streamed once) and everything between them;
 * encoder.py hands larger batches' products to the library GEMM (hipBLASLt), which wins there.  All tensors are float32, row-major, device
 * pointers owned by the caller; functions are stateless (no engine handle), enqueue on `stream`, never synchronise, and
 * return 0 or a negative msr_status (msr_last_error(NULL) holds the text).
 *
 * Architecture restated from the reference's dependency (transformers `ModernBertModel`, sentence-transformers 5.0.0 in
 * requirements.txt:13): token embedding -> LayerNorm (no bias); 22 layers of
 *     h += Wo . attention(rope(Wqkv . norm(h)))      (layer 0: no norm; every third layer global attention with
 *                                                      rope theta 160000, the others a +-64 token window, theta 10000)
 *     h += Wo . (gelu(u[:1152]) * u[1152:]),  u = Wi . norm(h)
 * -> final LayerNorm -> mean over the tokens of each sequence. */
#ifndef MSRETR_ENCODER_H
#define MSRETR_ENCODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y[m] = LayerNorm(x[m]) * w over `dim` (768 or 1024) features, biased variance, no bias term.
 * x == NULL with ids / table given: x[m] = table[ids[m]] (embedding lookup fused in).  m in [0, n_rows). */
int msr_enc_layernorm(const float* x, const int32_t* ids, const float* table, const float* w, float* y,
                      int64_t n_rows, int32_t dim, float eps, void* stream);

/* Self-attention of short sequences.  qkv is [n_tok][3][n_heads][64] (the Wqkv product); sequence b owns tokens
 * [seq_off[b], seq_off[b+1]) (<= 128 each).  Rotary embedding (rotate-half form) with inv_freq[32] applied to q and k,
 * scores q.k / 8, keys farther than `window` positions away masked out (window <= 0: none), softmax, times v.
 * max_len: an upper bound of the sequence lengths the caller vouches for (<= 128; 0 = no better bound than 128): up to 32
 * a WAVE serves a (sequence, head) pair instead of a workgroup (a batch of queries: 1536 pairs of 8 tokens); a sequence
 * longer than the bound gets NaNs.  out is [n_tok][n_heads * 64]. */
int msr_enc_attention(const float* qkv, const int32_t* seq_off, int32_t n_seq, int32_t n_heads, const float* inv_freq,
                      int32_t window, int32_t max_len, float* out, void* stream);

/* y[m][j] = gelu(u[m][j]) * u[m][half + j], j < half (exact erf GELU). */
int msr_enc_geglu(const float* u, float* y, int64_t n_rows, int32_t half, void* stream);

/* out[b] = mean of h[seq_off[b] .. seq_off[b+1]) (all-zero for an empty sequence); normalize != 0: divided by its
 * L2 norm afterwards (sentence-transformers normalize_embeddings). */
int msr_enc_mean_pool(const float* h, const int32_t* seq_off, int32_t n_seq, int32_t dim, int32_t normalize, float* out,
                      void* stream);

/* y[t][o] = sum_i x[t][i] * w[o][i] (+ resid[t][o] when resid != NULL; resid may alias y): torch.nn.Linear without bias,
 * the form of every projection of the model (Wqkv 2304x768, attention Wo 768x768, mlp Wi 2304x768, mlp Wo 768x1152).
 * x is [n_tok][n_in], w [n_out][n_in] (the checkpoint's layout), y / resid [n_tok][n_out]; n_out % 32 == 0,
 * n_in % 128 == 0, pointers 16-byte aligned.  Exact f32 products with f32 accumulation (v_mfma_f32_16x16x4_f32); the
 * summation order is fixed, so results are reproducible run to run. */
int msr_enc_linear(const float* x, const float* w, const float* resid, float* y, int32_t n_tok, int32_t n_out,
                   int32_t n_in, void* stream);

#ifdef __cplusplus
}
#endif
#endif
