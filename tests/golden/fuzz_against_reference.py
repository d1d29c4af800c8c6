#!/usr/bin/env python3
"""Randomised differential run of the ORACLE (and of the product's host path with the oracle as device backend) against the
REFERENCE'S OWN CODE, live — the same construction as make_golden.py (which freezes five such runs into fixtures), over as
many drawn cases as the time budget allows.  Build container only: it imports /root/reference.

    python tests/golden/fuzz_against_reference.py --seconds 300 [--seed 1]

Per case, the reference's real ``llama_duo_attention_forward_one_way_reordered_static`` (llama.py:309-434) over its real
``DuoAttentionStaticKVCache`` (static_kv_cache.py:18-315) — random ragged head split per layer, GQA group, sink / recent, RoPE
base and factor, batch rows with per-row position offsets, chunk lengths, decode steps with ``evict_last(1)`` — and its
tuple-cache forward ``llama_duo_attention_forward_one_way_reordered`` (llama.py:146-306) are run next to
  (a) ``oracle.duo_oracle.static_forward_ref`` / ``tuple_forward_ref``  and
  (b) the product's ``duo_static_attention_core`` / ``duo_attention_forward_one_way_reordered`` with the oracle backend,
on the same inputs: outputs equal up to one bf16 ulp on <= 2 % of the elements (the reference's stubs rotate in fp64 and use
torch SDPA; tests/test_oracle_golden.py::ulp_close), counters equal, V pools / tuple caches bit for bit, K pools to that ulp."""
import argparse
import os
import random
import sys
import time
import traceback
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "duo-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import make_golden as MG  # noqa: E402

D = 128


def ulp_close(ours, ref, what, max_frac=0.02):
    o, r = ours.float(), ref.float()
    assert o.shape == r.shape, (what, o.shape, r.shape)
    if o.numel() == 0:
        return
    diff = (o - r).abs()
    tol = torch.clamp(torch.maximum(o.abs(), r.abs()) * 2.0 ** -7, min=1e-3 * float(r.pow(2).mean().sqrt()))
    assert (diff <= tol).all(), f"{what}: max diff {diff.max():.3e} exceeds one bf16 ulp"
    frac = (diff > 0).float().mean().item()
    assert frac <= max(max_frac, 4.0 / o.numel()), f"{what}: {frac:.3%} of elements differ"


def draw_static(rng):
    Hkv = rng.choice([1, 2, 3, 4])
    group = rng.choice([1, 2, 3, 4])
    L = rng.choice([1, 2, 3])
    B = rng.choice([1, 1, 2])
    return dict(kind="static", Hkv=Hkv, group=group, counts=[rng.choice([0, Hkv, rng.randint(0, Hkv)]) for _ in range(L)],
                sink=rng.choice([1, 2, 4, 16]), recent=rng.choice([1, 3, 8, 32]), B=B,
                starts=[0] + [rng.randint(0, 9) for _ in range(B - 1)],
                chunks=[rng.randint(1, 70) for _ in range(rng.randint(1, 4))], decode_steps=rng.randint(0, 4),
                evict_n=rng.choice([1, 1, 2, 3]), second_prompt=[rng.randint(1, 40) for _ in range(rng.randint(0, 2))],
                theta=rng.choice([1e4, 5e5, 1e6]), factor=rng.choice([None, None, 2.0, 8.0]), seed=rng.randint(0, 2 ** 31 - 1))


def draw_tuple(rng):
    Hkv = rng.choice([1, 2, 4])
    return dict(kind="tuple", Hkv=Hkv, group=rng.choice([1, 2, 4]), nf=rng.randint(0, Hkv), sink=rng.choice([1, 4, 16]), B=rng.choice([1, 1, 2, 3]),
                recent=rng.choice([2, 8, 32]), steps=[rng.randint(1, 60) for _ in range(rng.randint(1, 3))] + [1] * rng.randint(0, 5),
                theta=rng.choice([1e4, 5e5]), seed=rng.randint(0, 2 ** 31 - 1))


def run_static(c):
    ref_fwd, RefCache = _REF["static_fwd"], _REF["Cache"]                                                  # THE REFERENCE
    ours = _ours()
    from oracle.duo_oracle import StaticCacheRef, static_forward_ref

    Hkv, G, counts, B = c["Hkv"], c["group"], c["counts"], c["B"]
    Hq, L = Hkv * G, len(counts)
    heads = [[1.0] * nf + [0.0] * (Hkv - nf) for nf in counts]
    total = max(sum(c["chunks"]) + c["decode_steps"], sum(c.get("second_prompt", []))) + 2
    model = types.SimpleNamespace(
        config=types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=Hq * D),
        parameters=lambda: iter([torch.zeros(1, dtype=torch.bfloat16)]))
    r_cache = RefCache(model, heads, B, total, c["sink"], c["recent"])
    attn = MG.fake_attention(Hq, Hkv, D, c["theta"], c["factor"])
    o_cache = StaticCacheRef(L, Hkv, D, heads, B, total, c["sink"], c["recent"])
    p_cache = ours["Cache"](ours["ShapeModel"](L, Hq, Hkv, D), heads, B, total, c["sink"], c["recent"])
    factor = 1.0 if c["factor"] is None else c["factor"]
    g = torch.Generator().manual_seed(c["seed"])
    pos = 0
    # chunked prefill, decode steps each followed by evict_last(n) (n = 1: the benchmark's protocol; n up to 3 rewinds into the
    # prompt), then — when drawn — clear() and a second prompt through the same caches
    steps = [(S, False) for S in c["chunks"]] + [(1, True)] * c["decode_steps"]
    if c.get("second_prompt"):
        steps += [("clear", False)] + [(S, False) for S in c["second_prompt"]]
    for si, (S, evict) in enumerate(steps):
        if S == "clear":
            for cache in (r_cache, o_cache, p_cache):
                cache.clear()
            pos = 0
            continue
        position_ids = torch.stack([torch.arange(pos + s0, pos + s0 + S) for s0 in c["starts"]])
        p0 = pos if B == 1 else [pos + s0 for s0 in c["starts"]]
        for l in range(L):
            h = torch.randn(B, S, Hq * D, generator=g).to(torch.bfloat16)
            split = lambda: (h.clone().view(B, S, Hq, D), h[..., : Hkv * D].clone().view(B, S, Hkv, D),
                             h[..., Hq * D - Hkv * D:].clone().view(B, S, Hkv, D))
            want, _ = ref_fwd(attn, h.clone(), position_ids=position_ids, kv_cache=r_cache, layer_idx=l)
            what = f"step {si} (S={S}) layer {l} pos {pos}"
            got_o = static_forward_ref(*split(), o_cache, l, p0, factor, c["theta"], round_p=False)
            ulp_close(got_o.reshape(B, S, Hq * D), want, what + ": oracle output")
            got_p = ours["core"](*split(), p_cache, l, p0, factor, c["theta"])
            ulp_close(got_p.reshape(B, S, Hq * D), want, what + ": product host path output")
            for name, cache in (("oracle", o_cache), ("product", p_cache)):
                n, m = r_cache.kv_seq_len_list[l], r_cache.streaming_kv_seq_len_list[l]
                assert (cache.kv_seq_len_list[l], cache.streaming_kv_seq_len_list[l]) == (n, m), f"{what}: {name} counters"
                ulp_close(cache.full_key_states_list[l][:, :n], r_cache.full_key_states_list[l][:, :n], f"{what}: {name} full K pool")
                assert torch.equal(cache.full_value_states_list[l][:, :n], r_cache.full_value_states_list[l][:, :n]), f"{what}: {name} full V pool"
                ulp_close(cache.streaming_key_states_list[l][:, :m], r_cache.streaming_key_states_list[l][:, :m], f"{what}: {name} stream K pool")
                assert torch.equal(cache.streaming_value_states_list[l][:, :m], r_cache.streaming_value_states_list[l][:, :m]), f"{what}: {name} stream V pool"
        if evict:
            n_ev = c.get("evict_n", 1)
            for cache in (r_cache, o_cache, p_cache):
                cache.evict_last(n_ev)
            pos = pos + 1 - n_ev if pos + 1 - n_ev > 0 else 0
            if r_cache.kv_seq_len == 0:
                break       # (rewound to an empty cache: the next call would be a first chunk at position 0 — nothing new)
        else:
            pos += S
    for cache in (o_cache, p_cache):
        assert cache.kv_seq_len_list == r_cache.kv_seq_len_list and cache.streaming_kv_seq_len_list == r_cache.streaming_kv_seq_len_list


def run_tuple(c):
    ref_fwd = _REF["tuple_fwd"]                                                                              # THE REFERENCE
    ours = _ours()
    from oracle.duo_oracle import tuple_forward_ref

    Hkv, G, nf = c["Hkv"], c["group"], c["nf"]
    Hq = Hkv * G
    inv_freq = 1.0 / (c["theta"] ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))

    def rotary_emb(x, position_ids):
        freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos().to(x.dtype), emb.sin().to(x.dtype)

    def module():
        m = MG.fake_attention(Hq, Hkv, D, c["theta"], None)
        m.rotary_emb = rotary_emb
        m.sink_size, m.recent_size = c["sink"], c["recent"]
        m.register_buffer("full_attention_heads", torch.tensor([1.0] * nf + [0.0] * (Hkv - nf)))
        return m

    r_mod, p_mod = module(), module()
    p_mod.config = types.SimpleNamespace(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=Hq * D, rope_scaling=None)
    B = c.get("B", 1)
    g = torch.Generator().manual_seed(c["seed"])
    r_past, o_past, p_past, pos = None, None, None, 0
    for si, S in enumerate(c["steps"]):
        h = torch.randn(B, S, Hq * D, generator=g).to(torch.bfloat16)
        pid = torch.arange(pos, pos + S)[None].expand(B, S)
        want, _, r_past = ref_fwd(r_mod, h.clone(), position_ids=pid, past_key_value=r_past, use_cache=True)
        cos, sin = rotary_emb(h, pid)
        q = h.clone().view(B, S, Hq, D)
        k = h[..., : Hkv * D].clone().view(B, S, Hkv, D)
        v = h[..., Hq * D - Hkv * D:].clone().view(B, S, Hkv, D)
        q, k = ours["hf_rotary"](q, k, cos, sin, unsqueeze_dim=2)
        got_o, o_past = tuple_forward_ref(q, k, v, o_past, nf, c["sink"], c["recent"], round_p=False)
        what = f"step {si} (S={S}) pos {pos}"
        ulp_close(got_o.reshape(B, S, Hq * D), want, what + ": oracle output")
        got_p, _, p_past = ours["tuple_fwd"](p_mod, h.clone(), past_key_value=p_past, use_cache=True, position_embeddings=(cos, sin))
        ulp_close(got_p.reshape(B, S, Hq * D), want, what + ": product host path output")
        for name, past in (("oracle", o_past), ("product", p_past)):
            assert torch.equal(past[0], r_past[0]), f"{what}: {name} retrieval cache"
            assert torch.equal(past[1], r_past[1]), f"{what}: {name} streaming cache"
        pos += S
    ours["release"](p_mod)


def draw_utils(rng):
    return dict(kind="utils", L=rng.randint(1, 6), H=rng.choice([1, 2, 4, 8]), ties=rng.random() < 0.5,
                sparsity=rng.choice([None, 0.0, 1.0, -0.3, 1.4, rng.random(), rng.random()]), threshold=rng.choice([None, 0.5, rng.random()]),
                repeat=rng.choice([1, 2, 128]), channel=rng.choice(["in", "out"]), bias=rng.random() < 0.4, seed=rng.randint(0, 2 ** 31 - 1))


def run_utils(c):
    """sparsify_attention_heads (utils.py:353-373: same numpy random stream, same result or the same exception type),
    reorder_linear_weights / reorder_full_attn_heads (patch/utils.py:7-45)"""
    import numpy as np
    ref_rh, ref_rw, ref_sp = _REF["reorder_h"], _REF["reorder_w"], _REF["sparsify"]                          # THE REFERENCE
    ours = _ours()
    rs = np.random.RandomState(c["seed"] % (2 ** 32))
    heads = rs.rand(c["L"], c["H"])
    if c["ties"]:
        heads = np.round(heads, 1)

    def call(fn):
        np.random.seed(c["seed"] % (2 ** 32))
        try:
            return fn(heads.copy(), threshold=c["threshold"], sparsity=c["sparsity"])
        except Exception as e:      # noqa: BLE001
            return type(e)

    want, got = call(ref_sp), call(ours["sparsify"])
    if isinstance(want, type) and want is TypeError and c["sparsity"] is None and c["threshold"] is not None:
        # deliberate: utils.py:364-369 compares sparsity to 1 and 0 unconditionally, so the reference's threshold-only mode
        # raises TypeError; here it works (duo_attn/utils.py::sparsify_attention_heads says so)
        assert not isinstance(got, type)
        KNOWN["sparsify threshold-only mode works here, raises TypeError in the reference"] += 1
    elif isinstance(want, type):
        assert got is want, f"sparsify: the reference raises {want.__name__}, this package {got}"
    else:
        assert not isinstance(got, type), f"sparsify: this package raises {got}, the reference does not"
        assert np.array_equal(want[0], got[0]) and want[1] == got[1], "sparsify: result differs"
    pattern = torch.tensor((heads[0] > 0.5).astype(np.float32))
    n = c["H"] * c["repeat"]
    torch.manual_seed(c["seed"])
    lin = torch.nn.Linear(n if c["channel"] == "in" else 24, 24 if c["channel"] == "in" else n, bias=c["bias"])
    import copy

    a, b = copy.deepcopy(lin), copy.deepcopy(lin)
    def rw(fn, m):
        try:
            fn(m, pattern.clone(), c["repeat"], c["channel"])
        except Exception as e:      # noqa: BLE001  (the reference indexes the bias with the IN-channel mask: IndexError / wrong size)
            return type(e)
        return None

    ea, eb = rw(ref_rw, a), rw(ours["reorder_w"], b)
    if ea is not None and eb is None and c["channel"] == "in" and c["bias"]:
        # deliberate: patch/utils.py:27-32 indexes the (output-sized) bias with the IN-channel mask; a biased o_proj would
        # fail there.  Here the bias of an "in"-reordered module stays as it is (the output rows do not move).
        assert torch.equal(b.bias, lin.bias)
        KNOWN["reorder_linear_weights('in') with a bias works here, raises in the reference"] += 1
    else:
        assert ea is eb, f"reorder_linear_weights: the reference raises {ea}, this package {eb}"
    if ea is None:
        assert torch.equal(a.weight, b.weight) and (a.bias is None or torch.equal(a.bias, b.bias)), "reorder_linear_weights differs"
    assert torch.equal(ref_rh(pattern.clone()), ours["reorder_h"](pattern.clone())), "reorder_full_attn_heads differs"


def draw_layer(rng):
    Hkv = rng.choice([1, 2, 4])
    return dict(kind="layer", Hkv=Hkv, group=rng.choice([1, 2, 4] if Hkv < 4 else [1, 2]), nf=rng.randint(0, Hkv), inter=8 * rng.randint(4, 96),
                sink=rng.choice([2, 4, 16]), recent=rng.choice([4, 8, 32]), chunks=[rng.randint(1, 40) for _ in range(rng.randint(1, 3))],
                decode_steps=rng.randint(1, 5), theta=rng.choice([1e4, 5e5]), fused=rng.random() < 0.5, seed=rng.randint(0, 2 ** 31 - 1))


class _FusedOracleBackend:
    """the oracle backend + ``token_linear`` (the oracle's module-by-module restatement): lets ``duo_decode_layer_fused`` —
    on the GPU the HIP token-row linears — run on the CPU"""

    def __init__(self):
        from oracle.duo_oracle import OracleBackend

        self._inner = OracleBackend(round_p=False)

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def token_linear_fits(self, n_rows, n_in):
        return n_rows <= 4 and n_in % 8 == 0

    def token_linear(self, x, blocks, norm=None, x2=None, residual=None, norm_hf=False):
        from oracle.duo_oracle import token_linear_ref

        return token_linear_ref(x, blocks, norm=norm, x2=x2, residual=residual, norm_hf=norm_hf)


def run_layer(c):
    """a whole DECODER LAYER: the reference's ``duo_attn_static_kv_cache_llama_decoder_layer_forward`` (static_kv_cache.py:
    507-546) around its static attention forward, real nn.Linear / LlamaMLP / LlamaRMSNorm modules with its
    ``flashinfer_rmsnorm_forward``, its real cache — against this package's decoder-layer forward on the same weights, module
    by module or with the decode steps in the fused form (``duo_decode_layer_fused``; on those steps the reference's linears sum
    in fp64 like the fused form's CPU stand-in, see below): every hidden state within two bf16 ulps, >= 99 % of the elements
    bit-equal (tests/test_token_linear_cpu.py::_layer_close), cache V pools bit for bit."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaMLP, LlamaRMSNorm

    from duo_attn import backend

    ours = _ours()
    Hkv, G, nf = c["Hkv"], c["group"], c["nf"]
    Hq, H = Hkv * G, Hkv * G * D
    g = torch.Generator().manual_seed(c["seed"])
    heads = [[1.0] * nf + [0.0] * (Hkv - nf)]
    total = sum(c["chunks"]) + c["decode_steps"] + 2
    rw = lambda o, i: (torch.randn(o, i, generator=g) * i ** -0.5).to(torch.bfloat16)
    W = {nm: rw(o, i) for nm, o, i in (("q_proj", H, H), ("k_proj", Hkv * D, H), ("v_proj", Hkv * D, H), ("o_proj", H, H),
                                       ("gate_proj", c["inter"], H), ("up_proj", c["inter"], H), ("down_proj", H, c["inter"]))}
    NW = {nm: (torch.rand(H, generator=g) + 0.5).to(torch.bfloat16) for nm in ("input_layernorm", "post_attention_layernorm")}

    def build(attn, attn_fwd, layer_fwd, norm_fwd):
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin = torch.nn.Linear(W[nm].shape[1], W[nm].shape[0], bias=False).to(torch.bfloat16)
            lin.weight.data.copy_(W[nm])
            setattr(attn, nm, lin)
        attn.forward = types.MethodType(attn_fwd, attn)
        layer = torch.nn.Module()
        layer.self_attn = attn
        layer.mlp = LlamaMLP(LlamaConfig(hidden_size=H, intermediate_size=c["inter"])).to(torch.bfloat16)
        for nm in ("gate_proj", "up_proj", "down_proj"):
            getattr(layer.mlp, nm).weight.data.copy_(W[nm])
        for nm in NW:
            ln = LlamaRMSNorm(H, eps=1e-5).to(torch.bfloat16)
            ln.weight.data.copy_(NW[nm])
            ln.forward = types.MethodType(norm_fwd, ln)
            setattr(layer, nm, ln)
        layer.forward = types.MethodType(layer_fwd, layer)
        return layer.eval()

    model = types.SimpleNamespace(
        config=types.SimpleNamespace(num_hidden_layers=1, num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=H),
        parameters=lambda: iter([torch.zeros(1, dtype=torch.bfloat16)]))
    r_cache = _REF["Cache"](model, heads, 1, total, c["sink"], c["recent"])
    r_layer = build(MG.fake_attention(Hq, Hkv, D, c["theta"], None), _REF["static_fwd"], _REF["layer_fwd"], _REF["rmsnorm_fwd"])
    p_attn = torch.nn.Module()
    p_attn.config = types.SimpleNamespace(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=H, head_dim=D,
                                          rope_theta=c["theta"], rope_scaling=None)
    p_attn.head_dim = D
    p_layer = build(p_attn, ours["static_attn_fwd"], ours["layer_fwd"], ours["rmsnorm_fwd"])
    p_cache = ours["Cache"](ours["ShapeModel"](1, Hq, Hkv, D), heads, 1, total, c["sink"], c["recent"])
    duo = ours["duo"]
    if c["fused"]:
        backend._set_backend_for_testing(_FusedOracleBackend())
    # This run is about the module SEQUENCE around the attention op, so the reference's attention call gets the oracle's
    # arithmetic for its duration (the static runs above compare the attention itself, against make_golden's independent SDPA
    # stub): a hidden state that differs is then a difference in norm / projection / residual / MLP order or rounding.
    from oracle.duo_oracle import flash_attn_func_ref

    r_llama = _REF["llama_mod"]
    sdpa_stub = r_llama.flash_attn_func
    r_llama.flash_attn_func = lambda q, k, v, causal=True, dropout_p=0.0, **kw: flash_attn_func_ref(q, k, v, causal=causal, round_p=False)
    try:
        pos = 0
        for si, S in enumerate(list(c["chunks"]) + [1] * c["decode_steps"]):
            pid = torch.arange(pos, pos + S)[None]
            h = torch.randn(1, S, H, generator=g).to(torch.bfloat16)
            fused_step = c["fused"] and S == 1 and p_cache.kv_seq_len_list[0] > 0
            if fused_step:
                # The fused form differs from the module sequence ONLY in how a projection's dot products are summed (the HIP
                # kernel: fp32 in its own order; its CPU stand-in token_linear_ref: fp64) — so for these steps the reference's
                # nn.Linear modules sum in fp64 too, and the comparison is exact again: any difference left is a difference in
                # the module sequence, the roundings between the modules or the cache handling.
                lin_fwd = lambda self, x: (x.double() @ self.weight.double().t()).float().to(x.dtype)
                mods = [m_ for m_ in r_layer.modules() if isinstance(m_, torch.nn.Linear)]
                for m_ in mods:
                    m_.forward = types.MethodType(lin_fwd, m_)
            try:
                want = r_layer(h.clone(), position_ids=pid, kv_cache=r_cache, layer_idx=0)[0]
            finally:
                if fused_step:
                    for m_ in mods:
                        del m_.forward
            if fused_step:
                got = duo.duo_decode_layer_fused(p_layer, h.clone(), p_cache, 0, None, pid)
            else:
                got = p_layer(h.clone(), position_ids=pid, kv_cache=p_cache, layer_idx=0)[0]
            what = f"step {si} (S={S}{', fused' if c['fused'] and S == 1 else ''}) pos {pos}"
            o, r = got.float(), want.float()
            assert o.shape == r.shape, (what, o.shape, r.shape)
            diff = (o - r).abs()
            tol = torch.clamp(torch.maximum(o.abs(), r.abs()) * 2.0 ** -6, min=2e-3 * float(r.pow(2).mean().sqrt()))
            same = (diff == 0).float().mean().item()
            rel = float((o - r).norm() / r.norm().clamp_min(1e-9))
            assert (diff <= tol).all(), f"{what}: hidden state max diff {diff.max():.3e}, {int((diff > tol).sum())} elements beyond two ulps (rel L2 {rel:.2e}, {same:.3f} bit-equal)"
            assert same >= 0.99 or o.numel() < 400, f"{what}: only {same:.3f} of the hidden state bit-equal to the reference's"
            pos += S
        n, m = r_cache.kv_seq_len_list[0], r_cache.streaming_kv_seq_len_list[0]
        assert (p_cache.kv_seq_len_list[0], p_cache.streaming_kv_seq_len_list[0]) == (n, m), "counters"
        ulp_close(p_cache.full_value_states_list[0][:, :n], r_cache.full_value_states_list[0][:, :n], "full V pool", max_frac=0.01)
        ulp_close(p_cache.streaming_value_states_list[0][:, :m], r_cache.streaming_value_states_list[0][:, :m], "stream V pool", max_frac=0.01)
    finally:
        r_llama.flash_attn_func = sdpa_stub
        if c["fused"]:
            from oracle.duo_oracle import OracleBackend

            backend._set_backend_for_testing(OracleBackend(round_p=False))


def draw_model(rng):
    Hkv = rng.choice([1, 2, 4])
    L = rng.randint(1, 3)
    return dict(kind="model", family=rng.choice(["llama", "mistral"]), Hkv=Hkv, group=rng.choice([1, 2] if Hkv == 4 else [1, 2, 4]), L=L,
                inter=8 * rng.randint(4, 64), heads=[[float(rng.random() < 0.5) for _ in range(Hkv)] for _ in range(L)],
                sink=rng.choice([2, 4, 16]), recent=rng.choice([4, 8, 32]), B=rng.choice([1, 1, 2]),
                chunks=[rng.randint(1, 40) for _ in range(rng.randint(1, 3))], decode_steps=rng.randint(1, 4), evict=rng.random() < 0.5,
                path=rng.choice(["static", "static", "tuple", "full"]), explicit_positions=rng.random() < 0.3,
                starts=[rng.randint(0, 9) for _ in range(2)], seed=rng.randint(0, 2 ** 31 - 1))


def run_model(c):
    """A whole HuggingFace model through the reference's ENABLERS and model / decoder-layer / attention forwards —
    ``enable_{llama,mistral}_duo_attention_static_kv_cache_eval`` + ``DuoAttentionStaticKVCache`` driven like
    eval/efficiency/benchmark_static.py:68-105 (chunked prefill, decode, optional evict_last), or ``enable_duo_attention_eval``
    with tuple caches — against this package's same-named API on a copy of the same random-init model: the un-reordered
    pattern goes in, so the weight reordering (patch/utils.py:7-45, llama.py:523-546) is part of what is compared.  RoPE and
    attention arithmetic are the oracle's on both sides (see run_layer); every logit within two bf16 ulps, >= 99 % bit-equal."""
    import copy

    import numpy as np
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM

    from oracle.duo_oracle import flash_attn_func_ref

    ours = _ours()
    Hq = c["Hkv"] * c["group"]
    torch.manual_seed(c["seed"])
    kw = dict(hidden_size=Hq * D, intermediate_size=c["inter"], num_hidden_layers=c["L"], num_attention_heads=Hq,
              num_key_value_heads=c["Hkv"], head_dim=D, vocab_size=97, max_position_embeddings=2048, rope_theta=10000.0,
              attn_implementation="eager", tie_word_embeddings=False)
    base = (LlamaForCausalLM(LlamaConfig(**kw)) if c["family"] == "llama"
            else MistralForCausalLM(MistralConfig(sliding_window=None, **kw))).to(torch.bfloat16).eval()
    heads = np.array(c["heads"])
    B, total = c["B"], sum(c["chunks"]) + c["decode_steps"] + 2
    ids = torch.randint(0, 97, (B, sum(c["chunks"]) + c["decode_steps"]), generator=torch.Generator().manual_seed(c["seed"] ^ 1))

    def drive(model, make_cache):
        outs, pos = [], 0
        past = make_cache(model)
        for n in list(c["chunks"]) + [1] * c["decode_steps"]:
            kw_pos = {}
            if c.get("explicit_positions") and c["path"] == "static":
                # the caller's own position ids, one offset per batch row (a left-padded batch): the static forward hands
                # position_ids[:, 0] to the RoPE kernel (llama.py:347-352)
                kw_pos["position_ids"] = torch.stack([torch.arange(pos + s0, pos + s0 + n) for s0 in c["starts"][:B]])
            o = model(input_ids=ids[:, pos:pos + n], past_key_values=past, use_cache=True, **kw_pos)
            outs.append(o.logits[:, -1:].float())
            decode = pos >= sum(c["chunks"])
            if c["path"] in ("tuple", "full"):
                if not (decode and c["evict"]):
                    past = o.past_key_values
            elif decode and c["evict"]:
                past.evict_last(1)
            if not (decode and c["evict"]):
                pos += n
        return torch.cat(outs, 1)

    ref_model, our_model = copy.deepcopy(base), copy.deepcopy(base)
    r_mod = _REF[c["family"] + "_mod"]
    # names the reference reads from HF attention modules that transformers 5 dropped (environment shim, like make_golden's)
    for layer in ref_model.model.layers:
        a = layer.self_attn
        a.num_heads, a.num_key_value_heads, a.hidden_size, a.rope_theta = Hq, c["Hkv"], Hq * D, 10000.0
        a.num_key_value_groups = c["group"]
        a.rotary_emb = ref_model.model.rotary_emb
    sdpa_stub = r_mod.flash_attn_func
    # (flash-attn's own signature: flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, ...) — the tuple
    #  baseline passes dropout positionally, tuple_kv_cache.py:184-191)
    shared = lambda q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **kw_: flash_attn_func_ref(
        q, k, v, causal=causal, softmax_scale=softmax_scale, round_p=False)
    r_mod.flash_attn_func = shared
    r_tkv = _REF["tuple_kv_mod"]
    sdpa_stub_t, r_tkv.flash_attn_func = r_tkv.flash_attn_func, shared
    try:
        if c["path"] == "full":
            # the full-attention tuple baseline (tuple_kv_cache.py:38-120, :493-512 / :766-800): no DuoAttention at all
            r_tkv.enable_tuple_kv_cache(ref_model)
            want = drive(ref_model, lambda m: None)
        elif c["path"] == "static":
            getattr(r_mod, f"enable_{c['family']}_duo_attention_static_kv_cache_eval")(ref_model, heads.copy())
            want = drive(ref_model, lambda m: _REF["Cache"](m, heads, B, total, c["sink"], c["recent"]))
        else:
            getattr(r_mod, f"enable_{c['family']}_duo_attention_eval")(ref_model, heads.copy(), c["sink"], c["recent"])
            want = drive(ref_model, lambda m: None)
    finally:
        r_mod.flash_attn_func = sdpa_stub
        r_tkv.flash_attn_func = sdpa_stub_t
    if c["path"] == "full":
        ours["enable_tuple_full"](our_model)
        got = drive(our_model, lambda m: None)
    elif c["path"] == "static":
        ours["enable_static"][c["family"]](our_model, heads.copy())
        got = drive(our_model, lambda m: ours["Cache"](m, heads, B, total, c["sink"], c["recent"]))
    else:
        ours["enable_eval"](our_model, heads.copy(), c["sink"], c["recent"])
        got = drive(our_model, lambda m: None)
        ours["release"](our_model)
    diff = (got - want).abs()
    tol = torch.clamp(torch.maximum(got.abs(), want.abs()) * 2.0 ** -6, min=2e-3 * float(want.pow(2).mean().sqrt()))
    same = (diff == 0).float().mean().item()
    rel = float((got - want).norm() / want.norm().clamp_min(1e-9))
    assert (diff <= tol).all() and same >= 0.99, f"logits: max diff {diff.max():.3e}, rel L2 {rel:.2e}, {same:.4f} bit-equal to the reference's"


def draw_int4(rng):
    Hkv = rng.choice([1, 2, 4])
    L = rng.randint(1, 3)
    chunk = rng.choice([8, 16, 40])
    counts = [rng.choice([0, Hkv, rng.randint(0, Hkv)]) for _ in range(L)]
    if max(Hkv - nf for nf in counts) > max(counts):
        # the reference sizes its quantisation staging buffers by the LARGEST RETRIEVAL-head count of any layer
        # (int4_kv.py:226-241) and quantises the streaming heads through them too: a pattern whose streaming heads
        # outnumber that fails inside the reference's put().  Not drawn.
        counts[rng.randrange(L)] = Hkv
    return dict(kind="int4", Hkv=Hkv, group=rng.choice([1, 2, 4]), counts=counts,
                sink=rng.choice([2, 4, 16]), recent=rng.choice([4, 8, 32]), chunk=chunk,
                steps=[rng.randint(1, chunk) for _ in range(rng.randint(1, 5))] + [1] * rng.randint(0, 4),
                scale=rng.choice([0.3, 1.0, 5.0]), seed=rng.randint(0, 2 ** 31 - 1))


class _KernelStandIn:
    """``demo/quantize_int4.cu`` as the reference's int4_kv.py binds it (int4_kv.py:47-56, :80-86, :106-108), computed by the
    INT4 oracle — which tests/golden/int4_ref.npz pins bit for bit to a build of that very file"""

    @staticmethod
    def quantize_int4_with_zero_point_per_group(tensor, q_packed, scale, zero_point, group_size):
        import numpy as np
        from oracle.int4_oracle import quantize_int4_ref

        assert group_size == 128
        p, s_, z = quantize_int4_ref(tensor.float().numpy())
        q_packed.copy_(torch.from_numpy(p))
        scale.copy_(torch.from_numpy(s_)[..., None])
        zero_point.copy_(torch.from_numpy(z)[..., None])

    @staticmethod
    def dequantize_int4_with_zero_point_per_group(q_packed, scale, zero_point, group_size, buffer, N):
        from oracle.int4_oracle import dequantize_int4_ref

        d = dequantize_int4_ref(q_packed.numpy(), scale.reshape(-1).numpy(), zero_point.reshape(-1).numpy())
        buffer[: N * group_size].copy_(torch.from_numpy(d).reshape(-1))


class _HipStandIn:
    """the three INT4 data-movement entry points ``duo_attn/int4_kv.py`` calls, computed by the INT4 oracle on CPU tensors (the
    product itself has no CPU path; on the GPU these are HIP kernels pinned bit for bit to the same oracle, tests/test_int4*.py)"""

    @staticmethod
    def int4_quantize_batched(src, q_pool, sz_pool, row0):
        from oracle.int4_oracle import quantize_int4_ref

        if src.shape[1] == 0 or src.shape[2] == 0:
            return
        p, s_, z = quantize_int4_ref(src.float().numpy())
        n = src.shape[1]
        q_pool[:, row0:row0 + n].copy_(torch.from_numpy(p))
        sz_pool[:, row0:row0 + n, :, 0].copy_(torch.from_numpy(s_))
        sz_pool[:, row0:row0 + n, :, 1].copy_(torch.from_numpy(z))

    @staticmethod
    def int4_dequantize_batched(q_pool, sz_pool, n_tokens, out, fused=False):
        from oracle.int4_oracle import dequantize_int4_ref

        B, h = q_pool.shape[0], q_pool.shape[2]
        res = out[: B * n_tokens * h * 128].view(B, n_tokens, h, 128)
        if n_tokens and h and B:
            d = dequantize_int4_ref(q_pool[:, :n_tokens].contiguous().numpy(), sz_pool[:, :n_tokens, :, 0].contiguous().numpy(),
                                    sz_pool[:, :n_tokens, :, 1].contiguous().numpy(), fused=fused)
            res.copy_(torch.from_numpy(d))
        return res

    @staticmethod
    def int4_stream_compress_batched(kq, ksz, vq, vsz, length, sink, recent):
        for t in (kq, ksz, vq, vsz):
            if t.shape[2]:
                t[:, sink:sink + recent] = t[:, length - recent:length].clone()
        return sink + recent


def run_int4(c):
    """``DuoAttentionStaticINT4KVCache`` (demo/int4_kv.py:115-492, the REAL class, its CUDA extension replaced by the oracle's
    arithmetic) against this package's class of the same name with its three HIP entry points replaced likewise: random put /
    get / compress sequences as demo/w8a8kv4_llama.py:219-278 issues them — what ``put`` and ``get`` return, the counters, the
    packed pools and their (scale, zero) rows, bit for bit."""
    import numpy as np

    ours = _ours()
    Hkv, counts = c["Hkv"], c["counts"]
    Hq, L = Hkv * c["group"], len(counts)
    heads = [[1.0] * nf + [0.0] * (Hkv - nf) for nf in counts]
    total = sum(c["steps"]) + 2
    model = types.SimpleNamespace(
        config=types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=Hq * D),
        parameters=lambda: iter([torch.zeros(1, dtype=torch.float16)]))
    r = _REF["Int4Cache"](model, heads, 1, total, c["sink"], c["recent"], c["chunk"])
    p = ours["Int4Cache"](model, heads, 1, total, c["sink"], c["recent"], c["chunk"])
    g = torch.Generator().manual_seed(c["seed"])
    for si, S in enumerate(c["steps"]):
        for l, nf in enumerate(counts):
            k = (torch.randn(1, S, Hkv, D, generator=g) * c["scale"]).to(torch.float16)
            v = (torch.randn(1, S, Hkv, D, generator=g) * c["scale"]).to(torch.float16)
            what = f"step {si} (S={S}) layer {l}"
            want, got = r.put(l, k.clone(), v.clone()), p.put(l, k.clone(), v.clone())
            for name, a, b in zip(("full K", "full V", "stream K", "stream V"), want, got):
                assert a.numel() == b.numel() and torch.equal(a.reshape(-1), b.reshape(-1)), f"{what}: put() returns another {name}"
            assert r.kv_seq_len_list == p.kv_seq_len_list and r.streaming_kv_seq_len_list == p.streaming_kv_seq_len_list, what + ": counters after put"
            r.compress(l)
            p.compress(l)
            assert r.kv_seq_len_list == p.kv_seq_len_list and r.streaming_kv_seq_len_list == p.streaming_kv_seq_len_list, what + ": counters after compress"
            for name, a, b in zip(("full K", "full V", "stream K", "stream V"), r.get(l), p.get(l)):
                assert a.numel() == b.numel() and torch.equal(a.reshape(-1), b.reshape(-1)), f"{what}: get() after compress returns another {name}"
            n, m = r.kv_seq_len_list[l], r.streaming_kv_seq_len_list[l]
            for rc, pc, rows, nh in ((r.full_key_caches[l], p.full_key_caches[l], n, nf), (r.full_value_caches[l], p.full_value_caches[l], n, nf),
                                     (r.streaming_key_caches[l], p.streaming_key_caches[l], m, Hkv - nf),
                                     (r.streaming_value_caches[l], p.streaming_value_caches[l], m, Hkv - nf)):
                if nh:
                    assert torch.equal(rc.quantized_data[:, :rows], pc.quantized_data[:, :rows]), what + ": packed pool"
                    assert torch.equal(rc.scale[:, :rows], pc.scale[:, :rows]) and torch.equal(rc.zero_point[:, :rows], pc.zero_point[:, :rows]), what + ": scale / zero rows"
    assert r.kv_seq_len == p.kv_seq_len and r.streaming_kv_seq_len == p.streaming_kv_seq_len


_OURS, _REF = {}, {}
INDEPENDENT_ROPE = False
KNOWN = __import__("collections").Counter()


def _ours():
    return _OURS


def _load_both():
    """Both packages are called ``duo_attn``.  Ours is imported first and stays in sys.modules (its functions import lazily
    by relative name); the reference is imported while ours is hidden, its entry points are kept in ``_REF`` (it has no lazy
    imports), and its modules are taken out of sys.modules again."""
    import importlib

    import duo_attn.patch._duo as duo
    import duo_attn.patch.flashinfer_utils as fiu
    import duo_attn.patch.static_kv_cache as skv
    import duo_attn.patch.tuple_kv_cache as tkv
    import duo_attn.patch.utils as putils
    import duo_attn.utils as utils
    from duo_attn import backend
    from helpers import ShapeModel
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    _OURS.update(core=duo.duo_static_attention_core, Cache=skv.DuoAttentionStaticKVCache, ShapeModel=ShapeModel,
                 tuple_fwd=duo.duo_attention_forward_one_way_reordered, release=duo.release_tuple_arena,
                 hf_rotary=tkv.hf_apply_rotary_pos_emb, sparsify=utils.sparsify_attention_heads,
                 reorder_w=putils.reorder_linear_weights, reorder_h=putils.reorder_full_attn_heads,
                 static_attn_fwd=duo.duo_attention_forward_one_way_reordered_static,
                 layer_fwd=skv.duo_attn_static_kv_cache_decoder_layer_forward, rmsnorm_fwd=fiu.rmsnorm_forward, duo=duo)
    import duo_attn.patch as our_patch
    import duo_attn.patch.llama as our_llama
    import duo_attn.patch.mistral as our_mistral

    import duo_attn.int4_kv as our_int4

    our_int4._hip = _HipStandIn          # (the module's own name for the ctypes binding; the class is otherwise untouched)
    _OURS.update(Int4Cache=our_int4.DuoAttentionStaticINT4KVCache)
    _OURS.update(enable_static=dict(llama=our_llama.enable_llama_duo_attention_static_kv_cache_eval,
                                    mistral=our_mistral.enable_mistral_duo_attention_static_kv_cache_eval),
                 enable_eval=our_patch.enable_duo_attention_eval, enable_tuple_full=tkv.enable_tuple_kv_cache)
    is_pkg = lambda k: k == "duo_attn" or k.startswith("duo_attn.")
    mine = {k: v for k, v in sys.modules.items() if is_pkg(k)}
    for k in mine:
        del sys.modules[k]
    path0 = list(sys.path)
    sys.path[:] = [p for p in sys.path if not p.endswith("duo-attention_amd")]
    MG.install_shims()                      # puts /root/reference first on the path
    if not INDEPENDENT_ROPE:
        # make_golden's flashinfer stub rotates in fp64, the oracle (like the device) with an fp32 angle: the rotated bf16 rows
        # differ by one ulp on a fraction of a per cent of the elements, and over thousands of drawn cases a flipped key now
        # and then moves an output by two ulps.  RoPE is pinned on its own (K pools of the fixtures, tests/test_oracle_golden.py);
        # here the reference's forward gets the oracle's rotation, so that what is compared is the control flow, the cache and
        # the attention.  --independent-rope keeps the fp64 stub.
        from oracle.duo_oracle import rope_ref

        def apply_rope_inplace(q, k, indptr, offsets, interleave=False, rope_scale=1.0, rope_theta=1e4):
            assert not interleave
            for b in range(len(offsets)):
                lo, hi = int(indptr[b]), int(indptr[b + 1])
                for x in (q, k):
                    x[lo:hi] = rope_ref(x[lo:hi], int(offsets[b]), float(rope_scale), float(rope_theta))

        sys.modules["flashinfer"].rope = types.SimpleNamespace(apply_rope_inplace=apply_rope_inplace)
    importlib.invalidate_caches()
    import duo_attn as ref_pkg

    assert ref_pkg.__file__.startswith(MG.REF), ref_pkg.__file__
    import duo_attn.patch.flashinfer_utils as r_fiu
    import duo_attn.patch.llama as r_llama
    import duo_attn.patch.static_kv_cache as r_skv
    import duo_attn.patch.utils as r_putils
    import duo_attn.utils as r_utils

    _REF.update(static_fwd=r_llama.llama_duo_attention_forward_one_way_reordered_static, Cache=r_skv.DuoAttentionStaticKVCache,
                tuple_fwd=r_llama.llama_duo_attention_forward_one_way_reordered, sparsify=r_utils.sparsify_attention_heads,
                reorder_w=r_putils.reorder_linear_weights, reorder_h=r_putils.reorder_full_attn_heads,
                layer_fwd=r_skv.duo_attn_static_kv_cache_llama_decoder_layer_forward, rmsnorm_fwd=r_fiu.flashinfer_rmsnorm_forward,
                llama_mod=r_llama, mistral_mod=importlib.import_module("duo_attn.patch.mistral"),
                tuple_kv_mod=importlib.import_module("duo_attn.patch.tuple_kv_cache"))
    import importlib.util
    import torch.utils.cpp_extension as cpp_ext

    real_load = cpp_ext.load
    cpp_ext.load = lambda *a, **k: _KernelStandIn          # int4_kv.py:47-56 JIT-builds demo/quantize_int4.cu at import
    try:
        spec = importlib.util.spec_from_file_location("ref_demo_int4_kv", os.path.join(MG.REF, "demo", "int4_kv.py"))
        r_int4 = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(r_int4)
    finally:
        cpp_ext.load = real_load
    _REF.update(Int4Cache=r_int4.DuoAttentionStaticINT4KVCache)
    for k in [k for k in sys.modules if is_pkg(k)]:
        del sys.modules[k]
    sys.modules.update(mine)
    sys.path[:] = path0
    import duo_attn as mine_pkg

    assert "duo-attention_amd" in mine_pkg.__file__, mine_pkg.__file__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many drawn cases instead of after --seconds "
                    "(a fixed workload: the slice run by tests/ does not depend on how loaded the machine is)")
    ap.add_argument("--independent-rope", action="store_true", help="keep make_golden's fp64 RoPE stub inside the reference")
    a = ap.parse_args()
    global INDEPENDENT_ROPE
    INDEPENDENT_ROPE = a.independent_rope
    if not os.path.isdir(MG.REF):
        raise SystemExit("/root/reference is not here: this script runs in the build container only")
    _load_both()
    rng = random.Random(a.seed)
    t0, n, bad, kinds = time.time(), 0, 0, {"static": 0, "tuple": 0, "utils": 0, "layer": 0, "model": 0, "int4": 0}
    with torch.no_grad():
        while (n < a.cases) if a.cases > 0 else (time.time() - t0 < a.seconds):
            u = rng.random()
            c = (draw_static(rng) if u < 0.35 else draw_tuple(rng) if u < 0.58 else draw_utils(rng) if u < 0.64 else draw_layer(rng) if u < 0.76
                 else draw_model(rng) if u < 0.9 else draw_int4(rng))
            n += 1
            kinds[c["kind"]] += 1
            try:
                {"static": run_static, "tuple": run_tuple, "utils": run_utils, "layer": run_layer, "model": run_model, "int4": run_int4}[c["kind"]](c)
            except Exception as e:      # noqa: BLE001
                bad += 1
                print("FAIL", c, "\n    ", f"{type(e).__name__}: {str(e)[:600]}", flush=True)
                if not isinstance(e, AssertionError):
                    traceback.print_exc()
    for k, v in KNOWN.items():
        print(f"known, deliberate divergence x{v}: {k}")
    print(f"{n} cases ({kinds}) against the reference's own code in {time.time() - t0:.0f} s, {bad} failed (seed {a.seed})")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
