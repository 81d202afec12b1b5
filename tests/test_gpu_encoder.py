"""Query encoder on the GPU against transformers' ModernBertModel -- the reference's own dependency
(sentence-transformers 5.0.0, requirements.txt:13; reranker_api.py:137-139,355) -- on the SAME random weights: the served
checkpoint is fetched by name in the reference and is not available offline, so the architecture is pinned, the trained
weights are not ("parity unpinned" for them).  Container: transformers 5.15.0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def enc_world():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from transformers import ModernBertConfig, ModernBertModel
    from msretr.encoder import QueryEncoder
    torch.manual_seed(11)
    cfg = ModernBertConfig(reference_compile=False, attn_implementation="eager")
    hf = ModernBertModel(cfg).eval()
    with torch.no_grad():                                     # LayerNorm weights away from 1 so that they matter
        for n, p in hf.named_parameters():
            if n.endswith("norm.weight"):
                p.add_(0.2 * torch.randn_like(p))
    hf = hf.to("cuda")
    enc = QueryEncoder(hf.state_dict(), device=0)
    return hf, enc


def _hf_pooled(hf, seqs, pad_id=50283):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad_id, dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = torch.tensor(s)
        mask[i, :len(s)] = 1
    with torch.no_grad():
        h = hf(input_ids=ids.cuda(), attention_mask=mask.cuda()).last_hidden_state
    m = mask.cuda().unsqueeze(-1).to(h.dtype)
    return (h * m).sum(1) / m.sum(1).clamp(min=1)            # sentence-transformers mean pooling


def test_encoder_matches_transformers_modernbert(enc_world):
    hf, enc = enc_world
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 50000, size=n).tolist() for n in (1, 2, 5, 17, 64, 65, 100, 128, 9, 9)]
    got = enc.encode(seqs)
    ref = _hf_pooled(hf, seqs)
    err = float((got - ref).abs().max())
    print(f"max |encoder - transformers| over {len(seqs)} x 768 pooled values: {err:.3e}")
    assert got.shape == (len(seqs), 768) and err <= 2e-4
    # cosine between the two embeddings of every sequence: what the retriever consumes
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=1)
    assert float(cos.min()) > 1 - 1e-6
    # the batch above (400 tokens) runs msr_enc_linear's batch form (64 x 48 tiles); single queries its token-tile forms: one
    # per tile shape of that kernel (1, 17, 65, 128 tokens), same bar against transformers
    for i in (0, 3, 5, 7):
        single = enc.encode([seqs[i]])[0]
        e1 = float((single - ref[i]).abs().max())
        print(f"  {len(seqs[i])} tokens alone: max |encoder - transformers| = {e1:.3e}")
        assert e1 <= 2e-4 and float(torch.nn.functional.cosine_similarity(single, ref[i], dim=0)) > 1 - 1e-6
    # batching does not change a sequence's embedding (no padding token takes part in any product)
    one = enc.encode([seqs[4]])
    assert float((one[0] - got[4]).abs().max()) <= 1e-5
    # replaying the captured hipGraph == launching the kernels one by one, bit for bit; and a second call replays
    enc.use_graphs = False
    eager = enc.encode(seqs)
    enc.use_graphs = True
    assert torch.equal(eager, got) and torch.equal(enc.encode(seqs), got) and len(enc._graphs) >= 1
    nrm = enc.encode(seqs[:3], normalize=True)
    assert torch.allclose(nrm.norm(dim=1), torch.ones(3, device=nrm.device), atol=1e-5)
    assert torch.allclose(nrm, torch.nn.functional.normalize(got[:3], dim=1), atol=1e-6)


def test_encoder_kernels_against_torch_ops(enc_world):
    import ctypes as C
    _, enc = enc_world
    lib, dev = enc.lib, enc.device
    P = lambda t: C.c_void_p(t.data_ptr())
    S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    x = (torch.randn(37, 768, generator=g) * 3 + 0.5).to(dev)
    w = (1 + 0.3 * torch.randn(768, generator=g)).to(dev)
    y = torch.empty_like(x)
    assert lib.msr_enc_layernorm(P(x), None, None, P(w), P(y), 37, 768, C.c_float(1e-5), S) == 0
    assert float((y - torch.nn.functional.layer_norm(x, (768,), w, None, 1e-5)).abs().max()) <= 2e-6
    u = torch.randn(37, 2304, generator=g).to(dev) * 2
    a = torch.empty(37, 1152, device=dev)
    assert lib.msr_enc_geglu(P(u), P(a), 37, 1152, S) == 0
    assert float((a - torch.nn.functional.gelu(u[:, :1152]) * u[:, 1152:]).abs().max()) <= 2e-6
    off = torch.tensor([0, 5, 5, 37], dtype=torch.int32, device=dev)          # an empty sequence in the middle
    out = torch.empty(3, 768, device=dev)
    assert lib.msr_enc_mean_pool(P(x), P(off), 3, 768, 0, P(out), S) == 0
    assert float((out[0] - x[:5].mean(0)).abs().max()) <= 2e-6 and float(out[1].abs().max()) == 0.0
    assert float((out[2] - x[5:].mean(0)).abs().max()) <= 2e-6
    assert lib.msr_enc_layernorm(P(x), None, None, P(w), P(y), 37, 100, C.c_float(1e-5), S) < 0    # unsupported width
    assert b"dim=100" in lib.msr_last_error(None)


def test_encoder_linear_against_float64(enc_world):
    """msr_enc_linear (the skinny matrix product of every projection) against a float64 product: all four shapes of a
    layer, token counts on both sides of every tile size, residual in place, rows past the end untouched."""
    import ctypes as C
    _, enc = enc_world
    lib, dev = enc.lib, enc.device
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu"); g.manual_seed(23)
    for n_out, n_in in ((2304, 768), (768, 768), (768, 1152)):
        w = (torch.randn(n_out, n_in, generator=g) * 0.05).to(dev)
        for n_tok in (1, 5, 16, 17, 33, 64, 100, 128, 129, 300):
            x = torch.randn(n_tok, n_in, generator=g).to(dev)
            r = torch.randn(n_tok, n_out, generator=g).to(dev)
            want = x.double() @ w.double().t()
            y = torch.full((n_tok + 3, n_out), 7.0, device=dev)                    # 3 guard rows
            assert lib.msr_enc_linear(P(x), P(w), None, P(y), n_tok, n_out, n_in, S) == 0
            scale = float(want.abs().max())
            assert float((y[:n_tok].double() - want).abs().max()) <= 2e-6 * scale, (n_out, n_in, n_tok)
            assert bool((y[n_tok:] == 7.0).all())
            h = r.clone()
            assert lib.msr_enc_linear(P(x), P(w), P(h), P(h), n_tok, n_out, n_in, S) == 0     # h += x . w^T
            assert float((h.double() - (want + r.double())).abs().max()) <= 2e-6 * scale
            again = r.clone()
            assert lib.msr_enc_linear(P(x), P(w), P(again), P(again), n_tok, n_out, n_in, S) == 0
            assert torch.equal(h, again)                                           # fixed summation order
    x = torch.randn(4, 768, generator=g).to(dev); w = torch.randn(48, 768, generator=g).to(dev); y = torch.empty(4, 48, device=dev)
    assert lib.msr_enc_linear(P(x), P(w), None, P(y), 4, 48, 768, S) < 0 and b"n_out=48" in lib.msr_last_error(None)
    assert lib.msr_enc_linear(P(x), P(w), None, P(y), 4, 32, 100, S) < 0
    assert lib.msr_enc_linear(P(x), P(w), None, P(y), 0, 32, 768, S) == 0


def test_encoder_rejects_what_it_cannot_do(enc_world):
    _, enc = enc_world
    with pytest.raises(ValueError):
        enc.encode([[1] * 129])
    with pytest.raises(ValueError):
        enc.encode([[60000]])
    with pytest.raises(ValueError):
        enc.encode(["a query string"])                       # no tokenizer.json was loaded
    assert enc.encode([]).shape == (0, 768) and float(enc.encode([[]]).abs().max()) == 0.0


def test_encoder_from_local_directory_and_in_the_retriever(tmp_path):
    """from_dir: model.safetensors + tokenizer.json + modules.json of a local sentence-transformers directory (a 2-layer
    model and a word-level tokenizer written by this test), then the query STRING path of the host classes
    (reranker_api.py:355 -> Retriever.quick_search) with the encoder as the embedder."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from msretr.encoder import QueryEncoder, random_weights
    from msretr.retriever import Retriever
    from msretr.synthetic import synthetic_corpus
    d = tmp_path / "model"
    d.mkdir()
    w = random_weights(seed=4, layers=2)
    save_file({("model." + k): v.contiguous() for k, v in w.items()}, str(d / "model.safetensors"))
    vocab = {"[UNK]": 0, "tübingen": 1, "castle": 2, "food": 3, "and": 4, "drinks": 5}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(str(d / "tokenizer.json"))
    (d / "modules.json").write_text('[{"idx": 0, "type": "sentence_transformers.models.Transformer"}, '
                                    '{"idx": 1, "type": "sentence_transformers.models.Pooling"}, '
                                    '{"idx": 2, "type": "sentence_transformers.models.Normalize"}]')
    enc = QueryEncoder.from_dir(str(d), device=0)
    assert enc.layers == 2 and enc.normalize is True
    v = enc.encode("food and drinks")
    assert isinstance(v, np.ndarray) and v.shape == (768,) and abs(float(np.linalg.norm(v)) - 1.0) < 1e-5
    same = enc.encode([[3, 4, 5]], convert_to_numpy=True)[0]
    assert np.array_equal(v, same)                              # the string went through tokenizer.json
    assert not np.allclose(v, enc.encode("tübingen castle"))
    ix = synthetic_corpus(2000, n_chunks=8000, n_terms=1000, device="cuda")
    ix.urls = [f"https://example.org/{i}" for i in range(ix.n_docs)]
    ix.titles = [f"title {i}" for i in range(ix.n_docs)]
    ix.texts = [f"text {i}" for i in range(ix.n_docs)]
    r = Retriever(embedder=enc, indexer=ix)
    hits = r.quick_search("food and drinks", top_k=5, return_unique_docs=True)
    again = r.quick_search(None, top_k=5, query_embedding=v)
    assert len(hits) == 5 and [h["doc_id"] for h in hits] == [h["doc_id"] for h in again]
    with pytest.raises(FileNotFoundError):
        QueryEncoder.from_dir(str(tmp_path))
