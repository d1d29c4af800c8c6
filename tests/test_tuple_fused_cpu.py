"""CPU: the host logic of the tuple-cache decode step in its fused form (duo_attn/patch/_duo.py:
``duo_tuple_decode_layer_fused``) with the oracle behind it, and the oracle restatements it is checked with.

The product only takes the fused path on the GPU (``tuple_fused_decode_ok`` asks for CUDA tensors); here the decoder-layer
forward is wrapped so that q_len == 1 goes through it on the CPU, with the oracle's ``token_linear_ref`` /
``tuple_decode_prep_ref`` as the device backend, and the result is compared with the module-by-module tuple forward
(reference tuple_kv_cache.py:431-490 around llama.py:146-306) that ``tests/test_oracle_golden.py`` pins to the reference's own
outputs."""
import types

import numpy as np
import pytest
import torch

from oracle.duo_oracle import rmsnorm_hf_ref, rmsnorm_ref, token_linear_ref, tuple_decode_prep_ref
from test_token_linear_cpu import _FusedOracleBackend, _tiny_bf16


def test_rmsnorm_hf_ref_is_the_transformers_module():
    from transformers.models.llama.modeling_llama import LlamaRMSNorm

    torch.manual_seed(0)
    n = LlamaRMSNorm(256, eps=1e-5).to(torch.bfloat16)
    n.weight.data = (torch.rand(256) + 0.5).to(torch.bfloat16)
    x = torch.randn(3, 256).to(torch.bfloat16)
    want = n(x)
    assert torch.equal(rmsnorm_hf_ref(x, n.weight, 1e-5), want)
    # ... and it is NOT the flashinfer form (one rounding) the static path runs
    assert not torch.equal(rmsnorm_ref(x, n.weight, 1e-5), want)
    y = token_linear_ref(x, [(torch.eye(256).to(torch.bfloat16), None)], norm=(n.weight, 1e-5), norm_hf=True)
    assert torch.equal(y, want)


@pytest.mark.parametrize("nf,ns,n,sink,recent", [(1, 1, 5, 4, 12), (1, 1, 15, 4, 12), (1, 1, 16, 4, 12), (0, 2, 16, 4, 12),
                                                 (2, 0, 7, 4, 12), (1, 1, 0, 4, 12), (1, 2, 30, 4, 12), (1, 1, 16, 16, 0)])
def test_tuple_decode_prep_ref_is_the_reference_sequence(nf, ns, n, sink, recent):
    """``tuple_decode_prep_ref`` == the torch ops of the product's general tuple forward (itself golden-pinned to the
    reference's llama.py:146-306 in test_oracle_golden.py) on the same inputs: HF rotary, cat, truncation, stacking."""
    from duo_attn.patch.tuple_kv_cache import hf_apply_rotary_pos_emb

    g = torch.Generator().manual_seed(nf * 100 + ns * 10 + n)
    rn = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16)
    Hkv, G, D, N = nf + ns, 2, 128, 9
    q, k, v = rn(Hkv * G, D), rn(Hkv, D), rn(Hkv, D)
    ang = torch.rand(D // 2, generator=g) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()]).to(torch.bfloat16), torch.cat([ang.sin(), ang.sin()]).to(torch.bfloat16)
    arena = torch.zeros(2, nf, N + 4, D, dtype=torch.bfloat16)
    arena[:, :, :N] = rn(2, nf, N, D)
    before = arena.clone()
    old = rn(2, ns, n, D)
    # reference sequence (llama.py:177-184, :202-223, :273-301)
    qr, kr = hf_apply_rotary_pos_emb(q[None, None], k[None, None], cos[None, None], sin[None, None], unsqueeze_dim=2)
    sk = torch.cat([old[0].transpose(0, 1), kr[0, 0, nf:][None]], 0)
    sv = torch.cat([old[1].transpose(0, 1), v[nf:][None]], 0)
    if sk.shape[0] > sink + recent:
        sk = torch.cat([sk[:sink], sk[sk.shape[0] - recent:]], 0)[: sink + recent] if recent else sk[:sink]
        sv = torch.cat([sv[:sink], sv[sv.shape[0] - recent:]], 0)[: sink + recent] if recent else sv[:sink]
    want_stream = torch.stack([sk, sv], 0).transpose(1, 2)
    q2, k2 = q.clone(), k.clone()
    got = tuple_decode_prep_ref(q2, k2, v, cos, sin, nf, arena, N, old, sink, recent)
    assert torch.equal(q2, qr[0, 0]) and torch.equal(k2, kr[0, 0])
    assert got.shape == (2, ns, min(n + 1, sink + recent), D) and torch.equal(got, want_stream)
    assert torch.equal(arena[:, :, :N], before[:, :, :N]) and torch.equal(arena[:, :, N + 1:], before[:, :, N + 1:])
    if nf:
        assert torch.equal(arena[0, :, N], kr[0, 0, :nf]) and torch.equal(arena[1, :, N], v[:nf])


def _with_fused_tuple_layers(model, _duo):
    """q_len == 1 with a past goes through duo_tuple_decode_layer_fused (what the product's layer forward does on the GPU
    when tuple_fused_decode_ok says so)"""
    for layer in model.model.layers:
        orig = layer.forward

        def fwd(self, hidden_states, *a, _orig=orig, **kw):
            if hidden_states.shape[:2] == (1, 1) and kw.get("past_key_value") is not None:
                return _duo.duo_tuple_decode_layer_fused(self, hidden_states, kw["past_key_value"], kw["position_embeddings"])
            return _orig(hidden_states, *a, **kw)

        layer.forward = types.MethodType(fwd, layer)


@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_fused_tuple_decode_layer_host_logic_matches_module_by_module(family):
    from duo_attn import backend
    from duo_attn.patch import _duo, enable_duo_attention_eval

    heads = np.array([[1.0, 0.0], [0.0, 0.0], [1.0, 1.0]])           # layers with 1 / 0 / 2 retrieval kv heads of 2
    ids = torch.randint(0, 101, (1, 44), generator=torch.Generator().manual_seed(3))

    class Be(_FusedOracleBackend):
        def token_linear(self, x, blocks, norm=None, x2=None, residual=None, norm_hf=False):
            self.calls += 1
            self.hf += int(norm_hf)
            return token_linear_ref(x, blocks, norm=norm, x2=x2, residual=residual, norm_hf=norm_hf)

    be = Be()
    be.hf = 0
    backend._set_backend_for_testing(be)
    try:
        def run(fused):
            model = _tiny_bf16(family, 5)
            enable_duo_attention_eval(model, heads.copy(), 4, 12)      # window 16: slides during decode
            if fused:
                _with_fused_tuple_layers(model, _duo)
            outs, past = [], None
            with torch.no_grad():
                past = model(input_ids=ids[:, :10], past_key_values=None, use_cache=True).past_key_values
                for t in range(10, 44):
                    o = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
                    past = o.past_key_values
                    outs.append(o.logits.float())
            return torch.cat(outs, 1), past

        l_m, p_m = run(False)
        assert be.calls == 0
        l_f, p_f = run(True)
        assert be.calls == 34 * 3 * 4 and be.hf == 34 * 3 * 2        # both norms of every layer in HF's two-rounding form
    finally:
        backend._set_backend_for_testing(None)
    rel = ((l_f - l_m).norm() / l_m.norm()).item()
    assert rel < 2e-2, rel
    assert (l_f.argmax(-1) == l_m.argmax(-1)).float().mean() >= 0.9
    for l in range(3):
        nf = int(heads[l].sum())
        assert p_f[l][0].shape == p_m[l][0].shape == (2, nf, 44, 128)
        assert p_f[l][1].shape == p_m[l][1].shape == (2, 2 - nf, 16, 128)
        for a, b in zip(p_f[l], p_m[l]):
            if b.numel():
                assert ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-6)).item() < 2e-2
    # layer 0 sees the same inputs in both runs up to the first decode step: its first appended rows are bit-equal
    assert torch.equal(p_f[0][0][:, :, :10], p_m[0][0][:, :, :10])


def test_fused_tuple_decode_keeps_the_arena_contract():
    """the retrieval cache handed back is a view of the module's arena; an OLDER tuple re-used for a branch is copied into
    a fresh arena, so both continuations are right"""
    from duo_attn import backend
    from duo_attn.patch import _duo, enable_duo_attention_eval

    heads = np.array([[1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])
    ids = torch.randint(0, 101, (1, 20), generator=torch.Generator().manual_seed(4))
    backend._set_backend_for_testing(_FusedOracleBackend())
    try:
        model = _tiny_bf16("llama", 6)
        enable_duo_attention_eval(model, heads.copy(), 4, 12)
        _with_fused_tuple_layers(model, _duo)
        with torch.no_grad():
            p0 = model(input_ids=ids[:, :12], past_key_values=None, use_cache=True).past_key_values
            a1 = model(input_ids=ids[:, 12:13], past_key_values=p0, use_cache=True)
            arena = model.model.layers[0].self_attn._duo_full_kv_arena
            assert a1.past_key_values[0][0].data_ptr() == arena["buf"].data_ptr() and arena["len"] == 13
            a2 = model(input_ids=ids[:, 13:14], past_key_values=a1.past_key_values, use_cache=True)
            # branch: continue from p0 with another token, then replay the first continuation from p0 again
            b1 = model(input_ids=ids[:, 15:16], past_key_values=p0, use_cache=True)
            c1 = model(input_ids=ids[:, 12:13], past_key_values=p0, use_cache=True)
            c2 = model(input_ids=ids[:, 13:14], past_key_values=c1.past_key_values, use_cache=True)
        assert not torch.equal(b1.logits, a1.logits)
        assert torch.equal(c1.logits, a1.logits) and torch.equal(c2.logits, a2.logits)
        assert torch.equal(c2.past_key_values[0][0], a2.past_key_values[0][0])
        assert torch.equal(c2.past_key_values[0][1], a2.past_key_values[0][1])
    finally:
        backend._set_backend_for_testing(None)


def test_eligibility_is_refused_where_the_fused_form_does_not_apply():
    from duo_attn import backend
    from duo_attn.patch import _duo, enable_duo_attention_eval

    backend._set_backend_for_testing(_FusedOracleBackend())
    try:
        model = _tiny_bf16("llama", 7)
        enable_duo_attention_eval(model, np.array([[1.0, 0.0]] * 3), 4, 12)
        layer = model.model.layers[0]
        h = torch.zeros(1, 1, 256, dtype=torch.bfloat16)
        past = (torch.zeros(2, 1, 5, 128, dtype=torch.bfloat16), torch.zeros(2, 1, 5, 128, dtype=torch.bfloat16))
        pe = (torch.zeros(1, 1, 128, dtype=torch.bfloat16), torch.zeros(1, 1, 128, dtype=torch.bfloat16))
        # CPU tensors: never (the product has no CPU path); the oracle backend of the plain CPU suite has no tuple_decode_prep
        assert not _duo.tuple_fused_decode_ok(layer, h, past, pe, True)
        assert _duo._norm_form(layer.input_layernorm) == "hf"
        assert _duo._layer_static_verdict(layer, _duo.duo_attention_forward_one_way_reordered, "tuple") is False   # CPU weights
        # the static path's verdict is cached separately from the tuple path's
        assert _duo._layer_static_verdict(layer) is False
        assert set(layer._duo_fused_refs) == {"tuple", "static"}
    finally:
        backend._set_backend_for_testing(None)
