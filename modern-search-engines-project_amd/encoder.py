"""Query encoder on the GPU: the bi-encoder forward pass the reference runs through
sentence-transformers, `embedding_model.encode(request.query)` (reranker/reranker_api.py:137-139,355; index side
indexer/indexer.py:165 with normalize_embeddings=True).

The model is ModernBERT-base (embedder_training/train.py fine-tunes answerdotai/ModernBERT-base; the served checkpoint
`as-bessonov/reranker_searchengines_cos2` is fetched by NAME in the reference, which is not possible offline): weights come
from a LOCAL directory in the Hugging Face layout (`model.safetensors`, optional `tokenizer.json`, optional
sentence-transformers `modules.json`), or are random for tests and benchmarks.

For a query (<= 128 tokens in all) the whole forward pass is hand-written HIP behind the C ABI of
include/msretr_encoder.h: the matrix products (msr_enc_linear: skinny products on the exact-f32 matrix cores, each weight
read once) and everything between them -- embedding lookup + LayerNorm, LayerNorm, rotary embedding + attention, GeGLU,
masked mean pooling; torch owns the buffers and the hipGraph.  Batches take the same kernels (a tile shape of
msr_enc_linear for many tokens, the wave-per-(sequence, head) form of the attention for short sequences).  There is no CPU
fallback: without the library the class raises.  Parity: tests/test_gpu_encoder.py compares the output with transformers' ModernBertModel (the reference's
dependency) on the same random weights.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _abi

HIDDEN, HEADS, LAYERS, INTER, VOCAB = 768, 12, 22, 1152, 50368
GLOBAL_EVERY, LOCAL_WINDOW, THETA_GLOBAL, THETA_LOCAL, EPS = 3, 128, 160000.0, 10000.0, 1e-5
MAX_SEQ = 128                                               # msr_enc_attention: tokens per sequence


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def random_weights(seed=0, device="cpu", layers=LAYERS):
    """Random ModernBERT-base-shaped weights under the Hugging Face parameter names (tests, benchmarks)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    rnd = lambda *shape, std=0.02: (torch.randn(*shape, generator=g) * std).to(device)
    w = {"embeddings.tok_embeddings.weight": rnd(VOCAB, HIDDEN), "embeddings.norm.weight": 1 + rnd(HIDDEN, std=0.1),
         "final_norm.weight": 1 + rnd(HIDDEN, std=0.1)}
    for l in range(layers):
        p = f"layers.{l}."
        if l:
            w[p + "attn_norm.weight"] = 1 + rnd(HIDDEN, std=0.1)
        w[p + "attn.Wqkv.weight"] = rnd(3 * HIDDEN, HIDDEN)
        w[p + "attn.Wo.weight"] = rnd(HIDDEN, HIDDEN)
        w[p + "mlp_norm.weight"] = 1 + rnd(HIDDEN, std=0.1)
        w[p + "mlp.Wi.weight"] = rnd(2 * INTER, HIDDEN)
        w[p + "mlp.Wo.weight"] = rnd(HIDDEN, INTER)
    return w


class QueryEncoder:
    """encode(list of token-id lists | list of strings) -> float32 [n, 768] device tensor (mean-pooled, optionally
    L2-normalised).  Sequences are packed back to back (no padding tokens take part in any product)."""

    def __init__(self, weights, device=0, tokenizer=None, normalize=False, layers=None, use_graphs=True):
        self.lib = _abi.load()                               # raises if libmsretr.so is missing
        if not torch.cuda.is_available():
            raise _abi.MsrError(-101, "QueryEncoder needs a GPU: the encoder kernels have no CPU fallback")
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.tokenizer, self.normalize = tokenizer, bool(normalize)
        self.use_graphs, self._graphs = bool(use_graphs), {}
        strip = lambda k: k[len("model."):] if k.startswith("model.") else k
        self.w = {strip(k): v.to(device=self.device, dtype=torch.float32).contiguous() for k, v in weights.items()}
        self.layers = layers if layers is not None else 1 + max(int(k.split(".")[1]) for k in self.w if k.startswith("layers."))
        half = torch.arange(0, 64, 2, dtype=torch.int64).to(torch.float32) / 64.0
        self.inv_freq = {True: (1.0 / (THETA_GLOBAL ** half)).to(self.device), False: (1.0 / (THETA_LOCAL ** half)).to(self.device)}

    # ------------------------------------------------------------------ loading
    @staticmethod
    def from_dir(path, device=0):
        """Local Hugging Face / sentence-transformers directory: model.safetensors (+ tokenizer.json, modules.json)."""
        from safetensors.torch import load_file
        st = os.path.join(path, "model.safetensors")
        if not os.path.exists(st):
            raise FileNotFoundError(f"{st}: the encoder needs local weights (the reference fetches them by name)")
        weights = load_file(st)                              # safetensors: executes nothing from the file
        tok = None
        tj = os.path.join(path, "tokenizer.json")
        if os.path.exists(tj):
            from tokenizers import Tokenizer
            tok = Tokenizer.from_file(tj)
        normalize = False
        mj = os.path.join(path, "modules.json")
        if os.path.exists(mj):
            with open(mj, encoding="utf-8") as f:
                normalize = any("Normalize" in str(m.get("type", "")) for m in json.load(f))
        return QueryEncoder(weights, device=device, tokenizer=tok, normalize=normalize)

    # ------------------------------------------------------------------ forward
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.msr_last_error(None)
            raise _abi.MsrError(rc, msg.decode("utf-8", "replace") if msg else "?")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ln(self, x, weight, ids=None):
        n = int(ids.numel() if ids is not None else x.shape[0])
        y = torch.empty((n, HIDDEN), dtype=torch.float32, device=self.device)
        table = self.w["embeddings.tok_embeddings.weight"] if ids is not None else None
        self._check(self.lib.msr_enc_layernorm(_ptr(x), _ptr(ids), _ptr(table), _ptr(weight), _ptr(y), n, HIDDEN,
                                               C.c_float(EPS), self._stream()))
        return y

    def tokenize(self, texts):
        if self.tokenizer is None:
            raise ValueError("no tokenizer.json was loaded: pass token-id lists")
        return [enc.ids for enc in self.tokenizer.encode_batch(list(texts))]

    def encode(self, queries, normalize=None, convert_to_numpy=None):
        """A single string -> numpy float32 [768] (what SentenceTransformer.encode(str) returns, reranker_api.py:355);
        a list of strings or of token-id lists -> device tensor [n, 768] (numpy with convert_to_numpy=True)."""
        if isinstance(queries, str):
            v = self.encode([queries], normalize=normalize)[0]
            return v if convert_to_numpy is False else v.cpu().numpy()
        if convert_to_numpy:
            return self.encode(queries, normalize=normalize).cpu().numpy()
        seqs = self.tokenize(queries) if queries and isinstance(queries[0], str) else [list(map(int, q)) for q in queries]
        if any(len(s) > MAX_SEQ for s in seqs):
            raise ValueError(f"a sequence has more than {MAX_SEQ} tokens (query encoder: short texts only)")
        if any(t < 0 or t >= VOCAB for s in seqs for t in s):
            raise ValueError("token id outside the vocabulary")
        normalize = self.normalize if normalize is None else bool(normalize)
        off = np.zeros(len(seqs) + 1, np.int32)
        off[1:] = np.cumsum([len(s) for s in seqs])
        n_tok = int(off[-1])
        out = torch.zeros((len(seqs), HIDDEN), dtype=torch.float32, device=self.device)
        if n_tok == 0:
            return out
        ids = torch.tensor([t for s in seqs for t in s], dtype=torch.int32)
        seq_off = torch.from_numpy(off)
        max_len = max(len(s) for s in seqs)
        if not self.use_graphs:
            return self._forward(ids.to(self.device), seq_off.to(self.device), len(seqs), n_tok, normalize, out, max_len)
        # The forward pass is ~200 short launches (launch-bound for a query's few tokens): it is captured once per
        # (token count, sequence count, normalize) into a hipGraph over static buffers and replayed afterwards.
        max_len = 8 if max_len <= 8 else 16 if max_len <= 16 else 32 if max_len <= 32 else MAX_SEQ   # the attention kernel's classes
        key = (n_tok, len(seqs), bool(normalize), max_len)
        ent = self._graphs.get(key)
        if ent is None:
            s_ids = torch.zeros(n_tok, dtype=torch.int32, device=self.device)
            s_off = torch.zeros(len(seqs) + 1, dtype=torch.int32, device=self.device)
            s_out = torch.zeros((len(seqs), HIDDEN), dtype=torch.float32, device=self.device)
            s_ids.copy_(ids); s_off.copy_(seq_off)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                    # warm-up outside the capture (library handles, workspaces)
                self._forward(s_ids, s_off, len(seqs), n_tok, normalize, s_out, max_len)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._forward(s_ids, s_off, len(seqs), n_tok, normalize, s_out, max_len)
            if len(self._graphs) >= 64:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = (graph, s_ids, s_off, s_out)
        graph, s_ids, s_off, s_out = ent
        s_ids.copy_(ids, non_blocking=True); s_off.copy_(seq_off, non_blocking=True)
        graph.replay()
        return s_out.clone()

    def _linear(self, x, weight, y, resid=None):
        """y = x . weight^T (+ resid): msr_enc_linear, the HIP product on the exact-f32 matrix cores (K split over the waves of
        a workgroup; single queries: weights streamed once; batches: 64 x 48 tiles, one workgroup per CU).  There is no other
        implementation in the product; tools/encoder_bench.py swaps a library GEMM in for its comparison runs."""
        n_out, n_in = weight.shape
        self._check(self.lib.msr_enc_linear(_ptr(x), _ptr(weight), _ptr(resid), _ptr(y), int(x.shape[0]), int(n_out),
                                            int(n_in), self._stream()))
        return y

    def _forward(self, ids, seq_off, n_seq, n_tok, normalize, out, max_len=MAX_SEQ):
        w, st = self.w, self._stream
        h = self._ln(None, w["embeddings.norm.weight"], ids=ids)                      # lookup + LayerNorm
        new = lambda cols: torch.empty((n_tok, cols), dtype=torch.float32, device=self.device)
        qkv, att, u, act = new(3 * HIDDEN), new(HIDDEN), new(2 * INTER), new(INTER)
        for l in range(self.layers):
            p = f"layers.{l}."
            glob = l % GLOBAL_EVERY == 0
            x = h if l == 0 else self._ln(h, w[p + "attn_norm.weight"])               # layer 0 has no attn_norm
            self._linear(x, w[p + "attn.Wqkv.weight"], qkv)
            self._check(self.lib.msr_enc_attention(_ptr(qkv), _ptr(seq_off), n_seq, HEADS, _ptr(self.inv_freq[glob]),
                                                   0 if glob else LOCAL_WINDOW // 2, int(max_len), _ptr(att), st()))
            if l == 0:
                h = self._linear(att, w[p + "attn.Wo.weight"], new(HIDDEN), resid=h)  # (x is h in layer 0: keep it intact)
            else:
                self._linear(att, w[p + "attn.Wo.weight"], h, resid=h)                # h += att . Wo^T
            x = self._ln(h, w[p + "mlp_norm.weight"])
            self._linear(x, w[p + "mlp.Wi.weight"], u)
            self._check(self.lib.msr_enc_geglu(_ptr(u), _ptr(act), n_tok, INTER, st()))
            self._linear(act, w[p + "mlp.Wo.weight"], h, resid=h)                     # h += act . Wo^T
        h = self._ln(h, w["final_norm.weight"])
        self._check(self.lib.msr_enc_mean_pool(_ptr(h), _ptr(seq_off), n_seq, HIDDEN, int(normalize), _ptr(out), st()))
        return out
