"""CPU: ``token_linear_ref`` — the oracle of the fused decode-step linears (csrc/duo_linear.hip) — is the module sequence
it claims to be: an HF Llama decoder layer's non-attention half (reference static_kv_cache.py:482-537: input_layernorm,
q/k/v_proj, o_proj + residual, post_attention_layernorm, LlamaMLP, residual) evaluated module by module in bf16."""
import torch

from oracle.duo_oracle import rmsnorm_ref, token_linear_ref


def _ulp_close(a, b, what):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = (2.0 ** -7) * b.abs() + 1e-4
    assert (err <= tol).all(), f"{what}: worst {err.max().item():.3e}"
    assert (a == b).float().mean() >= 0.9, what


def test_ref_is_the_hf_module_sequence():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaMLP, LlamaRMSNorm

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    mlp = LlamaMLP(cfg).to(torch.bfloat16)
    ln = LlamaRMSNorm(256, eps=1e-5).to(torch.bfloat16)
    ln.weight.data = (torch.rand(256) + 0.5).to(torch.bfloat16)
    q = torch.nn.Linear(256, 256, bias=True).to(torch.bfloat16)
    k = torch.nn.Linear(256, 128, bias=False).to(torch.bfloat16)
    x = torch.randn(3, 256).to(torch.bfloat16)
    with torch.no_grad():
        # norm (HF's own RMSNorm rounds x*rs to the input dtype BEFORE the weight multiply; the static path swaps in the
        # flashinfer form — one rounding — which is what the oracle and the kernel implement: flashinfer_utils.py:9-26)
        xn = rmsnorm_ref(x, ln.weight, 1e-5)
        want_qk = torch.cat([q(xn), k(xn)], -1)
        got_qk = token_linear_ref(x, [(q.weight, q.bias), (k.weight, None)], norm=(ln.weight, 1e-5))
        _ulp_close(got_qk, want_qk, "norm + q|k")
        # o_proj-like product with the residual add
        want = x + q(x)
        got = token_linear_ref(x, [(q.weight, q.bias)], residual=x)
        _ulp_close(got, want, "linear + residual")
        # MLP: gate|up, then down over silu(g) * u, + residual
        gu = token_linear_ref(x, [(mlp.gate_proj.weight, None), (mlp.up_proj.weight, None)])
        got = token_linear_ref(gu[:, :704], [(mlp.down_proj.weight, None)], x2=gu[:, 704:], residual=x)
        want = x + mlp(x)
        _ulp_close(got, want, "mlp + residual")


def test_exact_returns_the_unrounded_product():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, generator=g).to(torch.bfloat16)
    w = torch.randn(5, 64, generator=g).to(torch.bfloat16)
    y, pre = token_linear_ref(x, [(w, None)], exact=True)
    assert pre.dtype == torch.float64 and torch.equal(y, pre.float().to(torch.bfloat16))
    assert torch.allclose(pre, x.double() @ w.double().t())


# ----------------------------------------------------------------------------- the fused decoder layer's host logic
class _FusedOracleBackend:
    """the oracle backend + ``token_linear`` (the oracle's module-by-module restatement): drives
    ``duo_decode_layer_fused`` on the CPU — the product itself only takes that path on the GPU"""

    def __init__(self):
        from oracle.duo_oracle import OracleBackend

        self._inner = OracleBackend(round_p=False)
        self.calls = 0

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def token_linear_fits(self, n_rows, n_in):
        return n_rows <= 4 and n_in % 8 == 0

    def token_linear(self, x, blocks, norm=None, x2=None, residual=None, norm_hf=False):
        self.calls += 1
        return token_linear_ref(x, blocks, norm=norm, x2=x2, residual=residual, norm_hf=norm_hf)


def _tiny_bf16(family, seed):
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM

    torch.manual_seed(seed)
    kw = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
              num_key_value_heads=2, head_dim=128, vocab_size=101, max_position_embeddings=1024,
              rope_theta=10000.0, attn_implementation="eager", tie_word_embeddings=False)
    m = LlamaForCausalLM(LlamaConfig(**kw)) if family == "llama" else MistralForCausalLM(MistralConfig(sliding_window=None, **kw))
    return m.to(torch.bfloat16).eval()


def _with_fused_layers(model, _duo):
    """wrap the patched decoder-layer forwards: q_len == 1 goes through duo_decode_layer_fused (what the product's
    layer forward does on the GPU when fused_decode_layer_ok says so)"""
    import types

    for layer in model.model.layers:
        orig = layer.forward

        def fwd(self, hidden_states, *a, _orig=orig, **kw):
            if hidden_states.shape[1] == 1 and kw.get("kv_cache") is not None and kw["kv_cache"].kv_seq_len_list[kw["layer_idx"]] > 0:
                assert _duo._layer_modules(self)[0] is self.self_attn.q_proj
                return (_duo.duo_decode_layer_fused(self, hidden_states, kw["kv_cache"], kw["layer_idx"], kw.get("pos0"),
                                                    kw.get("position_ids")),)
            return _orig(hidden_states, *a, **kw)

        layer.forward = types.MethodType(fwd, layer)


import pytest


@pytest.mark.parametrize("family,bsz", [("llama", 1), ("mistral", 2)])
def test_fused_decode_layer_host_logic_matches_module_by_module(family, bsz):
    """duo_decode_layer_fused (q|k|v with the norm folded in -> the static attention core on views of the fused buffer ->
    o_proj + residual -> gate|up with the norm folded in -> down_proj over silu(g)*u + residual) against the patched
    module-by-module layer forward, whole model, on the CPU with the oracle behind both: same tokens, logits within the
    bf16 noise of a different dot-product summation order; the cache ends in the same state"""
    import numpy as np

    from duo_attn import backend
    from duo_attn.patch import _duo

    mod = __import__(f"duo_attn.patch.{family}", fromlist=["x"])
    enable = getattr(mod, f"enable_{family}_duo_attention_static_kv_cache_eval")
    heads = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    ids = torch.randint(0, 101, (bsz, 40), generator=torch.Generator().manual_seed(3))
    be = _FusedOracleBackend()
    backend._set_backend_for_testing(be)
    try:
        def run(fused):
            model = _tiny_bf16(family, 5)
            enable(model, heads.copy())
            if fused:
                _with_fused_layers(model, _duo)
            cache = mod.DuoAttentionStaticKVCache(model, heads, bsz, 64, 4, 12)       # window 16: slides during decode
            outs = []
            with torch.no_grad():
                model(input_ids=ids[:, :30], past_key_values=cache, use_cache=True)
                for t in range(30, 40):
                    outs.append(model(input_ids=ids[:, t:t + 1], past_key_values=cache, use_cache=True).logits.float())
            return torch.cat(outs, 1), cache

        be.calls = 0
        l_m, c_m = run(False)
        assert be.calls == 0
        l_f, c_f = run(True)
        assert be.calls == 10 * 3 * 4                   # four token-row linears per layer and step
    finally:
        backend._set_backend_for_testing(None)
    rel = ((l_f - l_m).norm() / l_m.norm()).item()
    assert rel < 2e-2, rel
    assert (l_f.argmax(-1) == l_m.argmax(-1)).float().mean() >= 0.9
    assert c_f.kv_seq_len == c_m.kv_seq_len == 40
    for l in range(3):
        assert c_f.streaming_kv_seq_len_list[l] == c_m.streaming_kv_seq_len_list[l]
        a, b = c_f.full_value_states_list[l][:, :40].float(), c_m.full_value_states_list[l][:, :40].float()
        assert ((a - b).norm() / b.norm().clamp_min(1e-6)).item() < 2e-2


# ----------------------------------------------------------------------------- the reference's own decoder layer
def _bf16(a):
    import numpy as np

    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def _layer_from_golden(g, _duo):
    """the layer of tests/golden/layer_a.npz as the product sees an HF decoder layer: nn.Linear projections, HF's LlamaMLP and
    LlamaRMSNorm (patched RMSNorm forward), this package's static attention / decoder-layer forwards"""
    import types

    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaMLP, LlamaRMSNorm

    from duo_attn.patch.flashinfer_utils import rmsnorm_forward
    from duo_attn.patch.static_kv_cache import duo_attn_static_kv_cache_decoder_layer_forward

    Hq, Hkv, D, sink, recent, inter, nf = (int(x) for x in g["dims"])
    H = Hq * D
    attn = torch.nn.Module()
    attn.config = types.SimpleNamespace(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=H, head_dim=D,
                                        rope_theta=float(g["rope"][0]), rope_scaling=None)
    attn.head_dim = D
    for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
        w = _bf16(g[f"w_{nm}"])
        lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False).to(torch.bfloat16)
        lin.weight.data.copy_(w)
        setattr(attn, nm, lin)
    attn.forward = types.MethodType(_duo.duo_attention_forward_one_way_reordered_static, attn)
    layer = torch.nn.Module()
    layer.self_attn = attn
    layer.mlp = LlamaMLP(LlamaConfig(hidden_size=H, intermediate_size=inter)).to(torch.bfloat16)
    for nm in ("gate_proj", "up_proj", "down_proj"):
        getattr(layer.mlp, nm).weight.data.copy_(_bf16(g[f"w_{nm}"]))
    for nm in ("input_layernorm", "post_attention_layernorm"):
        ln = LlamaRMSNorm(H, eps=float(g["eps"])).to(torch.bfloat16)
        ln.weight.data.copy_(_bf16(g[f"w_{nm}"]))
        ln.forward = types.MethodType(rmsnorm_forward, ln)
        setattr(layer, nm, ln)
    layer.forward = types.MethodType(duo_attn_static_kv_cache_decoder_layer_forward, layer)
    return layer.eval(), (Hq, Hkv, D, sink, recent, nf)


def _layer_close(ours, ref, what):
    """bf16 layer output vs the reference's: the two sides differ by the attention stub's / the oracle's last bit and —
    fused form — by the summation order of the dot products; every element within two bf16 ulps of the value (or 2e-3 of
    the tensor's rms for near-zero outputs), at least 99 % bit-equal (measured on layer_a.npz: every element of every step
    bit-equal, module by module AND fused)"""
    o, r = ours.float(), ref.float()
    assert o.shape == r.shape, (what, o.shape, r.shape)
    diff = (o - r).abs()
    tol = torch.clamp(torch.maximum(o.abs(), r.abs()) * 2.0 ** -6, min=2e-3 * float(r.pow(2).mean().sqrt()))
    assert (diff <= tol).all(), f"{what}: max diff {diff.max():.3e}, {int((diff > tol).sum())} elements beyond the bar"
    same = (diff == 0).float().mean().item()
    assert same >= 0.99, f"{what}: only {same:.3f} of the elements bit-equal to the reference"
    return same


@pytest.mark.parametrize("fused", [False, True])
def test_decoder_layer_reproduces_the_reference_golden(fused):
    """tests/golden/layer_a.npz = the reference's OWN decoder-layer forward (static_kv_cache.py:507-546 around llama.py:
    309-434, flashinfer_utils.py:9-16) run on the CPU by tests/golden/make_golden.py.  This package's decoder layer —
    module by module, and with the decode steps in the fused form (duo_decode_layer_fused, the oracle's token_linear_ref
    standing in for the HIP kernel) — must reproduce its hidden states step by step and leave the cache in its state."""
    import os

    import numpy as np

    from duo_attn import backend
    from duo_attn.patch import _duo
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache
    from helpers import ShapeModel, heads_from_counts

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "layer_a.npz"))
    be = _FusedOracleBackend()
    backend._set_backend_for_testing(be)
    try:
        layer, (Hq, Hkv, D, sink, recent, nf) = _layer_from_golden(g, _duo)
        steps, n_pre = [int(x) for x in g["steps"]], int(g["n_prefill"])
        heads = heads_from_counts([nf], Hkv)
        cache = DuoAttentionStaticKVCache(ShapeModel(1, Hq, Hkv, D), heads, 1, sum(steps) + 2, sink, recent)
        pos, worst = 0, 1.0
        with torch.no_grad():
            for si, S in enumerate(steps):
                h = _bf16(g[f"h_{si}"])
                position_ids = torch.arange(pos, pos + S)[None]
                if fused and S == 1:
                    out = _duo.duo_decode_layer_fused(layer, h.clone(), cache, 0, pos, position_ids)
                else:
                    out = layer(h.clone(), position_ids=position_ids, kv_cache=cache, layer_idx=0, pos0=pos)[0]
                worst = min(worst, _layer_close(out, _bf16(g[f"o_{si}"]), f"layer_a step {si} (S={S}, fused={fused})"))
                pos += S
        assert be.calls == (4 * (len(steps) - n_pre) if fused else 0)
        n, m = (int(x) for x in g["len"])
        assert (cache.kv_seq_len_list[0], cache.streaming_kv_seq_len_list[0]) == (n, m)
        # V rows are projections of the normalised hidden state: bit-equal module by module, within the bar fused
        _layer_close(cache.full_value_states_list[0][:, :n], _bf16(g["fullv"]), "full V pool")
        _layer_close(cache.streaming_value_states_list[0][:, :m], _bf16(g["strv"]), "streaming V pool")
    finally:
        backend._set_backend_for_testing(None)
