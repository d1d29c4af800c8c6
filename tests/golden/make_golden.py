#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE'S OWN CODE (read-only at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

The reference cannot be imported as is: it targets transformers 4.45 and hard-imports CUDA-only
packages that are not installed (flash_attn, flashinfer, tensor_parallel).  This script
  * adds the few names the reference imports from HF modules that transformers 5.x dropped
    (typing aliases, CrossEntropyLoss),
  * stubs `tensor_parallel` (never exercised here),
  * stubs the two third-party ARITHMETIC entry points with restatements of their published
    semantics that are independent of this repo's oracle:
      flash_attn.flash_attn_func      -> torch.nn.functional.scaled_dot_product_attention with an
                                         explicit bottom-right-aligned causal mask, GQA by
                                         repeat_interleave, fp32 math (flash-attn 2.6.3 README:
                                         "causal mask aligned to the bottom right")
      flashinfer.rope.apply_rope_inplace -> rotate-half RoPE, fp64 angle = pos / rope_scale *
                                         theta^(-2i/D) (flashinfer docs, interleave=False)
and then calls the reference's real functions:
      duo_attn.utils.sparsify_attention_heads / load_attn_pattern      (utils.py:326-373)
      duo_attn.patch.utils.reorder_linear_weights / reorder_full_attn_heads (patch/utils.py:7-45)
      duo_attn.patch.static_kv_cache.DuoAttentionStaticKVCache          (static_kv_cache.py:18-315)
      duo_attn.patch.llama.llama_duo_attention_forward_one_way_reordered_static (llama.py:309-434)
      duo_attn.patch.llama.llama_duo_attention_forward_one_way_reordered        (llama.py:146-306)
      duo_attn.patch.static_kv_cache.duo_attn_static_kv_cache_llama_decoder_layer_forward (static_kv_cache.py:507-546)
      duo_attn.patch.flashinfer_utils.flashinfer_rmsnorm_forward                  (flashinfer_utils.py:9-16)
So the fixtures pin the reference's control flow, cache layout, head split/concat order and mask
alignment; the only thing they cannot pin is the last-bit behaviour of the CUDA kernels themselves.
"""
import os
import sys
import types
import typing

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_shims():
    import transformers.models.llama.modeling_llama as ml
    import transformers.models.mistral.modeling_mistral as mm

    for m in (ml, mm):
        for name in ("List", "Optional", "Tuple", "Union"):
            if not hasattr(m, name):
                setattr(m, name, getattr(typing, name))
        if not hasattr(m, "CrossEntropyLoss"):
            m.CrossEntropyLoss = torch.nn.CrossEntropyLoss

    def flash_attn_func(q, k, v, causal=True, dropout_p=0.0, **kw):
        B, Sq, Hq, D = q.shape
        Sk, Hkv = k.shape[1], k.shape[2]
        G = Hq // Hkv
        qf = q.float().transpose(1, 2)
        kf = k.float().transpose(1, 2).repeat_interleave(G, dim=1)
        vf = v.float().transpose(1, 2).repeat_interleave(G, dim=1)
        mask = None
        if causal:
            i = torch.arange(Sq)[:, None]
            j = torch.arange(Sk)[None, :]
            mask = j <= i + (Sk - Sq)
        o = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, attn_mask=mask)
        return o.transpose(1, 2).to(q.dtype)

    fa = types.ModuleType("flash_attn")
    fa.flash_attn_func = flash_attn_func
    fa.flash_attn_with_kvcache = None
    fa.flash_attn_varlen_func = None
    sys.modules["flash_attn"] = fa
    bp = types.ModuleType("flash_attn.bert_padding")
    bp.index_first_axis = bp.pad_input = bp.unpad_input = None
    sys.modules["flash_attn.bert_padding"] = bp

    class _Any(types.ModuleType):
        """stub package: any attribute is a dummy class (the TP code paths are never executed here)"""

        __path__ = []

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {})

    for sub in ("", ".config", ".communications", ".aux_actions", ".state_actions", ".pretrained_model",
                ".autoconfig"):
        sys.modules["tensor_parallel" + sub] = _Any("tensor_parallel" + sub)
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        sys.modules["matplotlib"] = _Any("matplotlib")
        sys.modules["matplotlib.pyplot"] = _Any("matplotlib.pyplot")

    def apply_rope_inplace(q, k, indptr, offsets, interleave=False, rope_scale=1.0, rope_theta=1e4):
        assert not interleave
        nnz, _, D = q.shape
        i = torch.arange(D // 2, dtype=torch.float64)
        freq = (1.0 / rope_scale) * torch.pow(torch.tensor(float(rope_theta), dtype=torch.float64), -2.0 * i / D)
        for b in range(len(offsets)):
            lo_, hi_ = int(indptr[b]), int(indptr[b + 1])
            pos = (int(offsets[b]) + torch.arange(hi_ - lo_, dtype=torch.float64))[:, None] * freq[None, :]
            cos, sin = torch.cos(pos)[:, None, :], torch.sin(pos)[:, None, :]
            for x in (q, k):
                xf = x[lo_:hi_].double()
                a, c = xf[..., : D // 2], xf[..., D // 2:]
                x[lo_:hi_] = torch.cat([a * cos - c * sin, c * cos + a * sin], dim=-1).to(x.dtype)

    def rmsnorm(x, w, eps=1e-6):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()).to(x.dtype)

    fi = types.ModuleType("flashinfer")
    fi.rope = types.SimpleNamespace(apply_rope_inplace=apply_rope_inplace)
    fi.norm = types.SimpleNamespace(rmsnorm=rmsnorm)
    sys.modules["flashinfer"] = fi
    sys.path.insert(0, REF)


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> uint16 numpy (lossless)."""
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


class Select(torch.nn.Module):
    """Deterministic stand-in for a projection: a column slice of the hidden state (no weights to ship)."""

    def __init__(self, lo, hi):
        super().__init__()
        self.lo, self.hi = lo, hi

    def forward(self, x):
        return x[..., self.lo:self.hi].clone()


def fake_attention(Hq, Hkv, D, rope_theta, rope_factor):
    m = torch.nn.Module()
    m.num_heads, m.num_key_value_heads, m.head_dim = Hq, Hkv, D
    m.num_key_value_groups = Hq // Hkv
    m.hidden_size = Hq * D
    m.rope_theta = rope_theta
    m.config = types.SimpleNamespace(rope_scaling=None if rope_factor is None else {"factor": rope_factor})
    m.q_proj = torch.nn.Identity()
    m.k_proj = Select(0, Hkv * D)
    m.v_proj = Select(Hq * D - Hkv * D, Hq * D)
    m.o_proj = torch.nn.Identity()
    return m


def golden_static(name, counts, Hq, Hkv, chunks, decode_steps, sink, recent, theta, factor, seed, batch=1, starts=None):
    """Chunked prefill + decode (with the benchmark's evict_last(1)) through the reference's static
    forward and its real DuoAttentionStaticKVCache, for every layer of a ragged head split.
    ``batch`` > 1: the reference's batch dimension (static_kv_cache.py:60-99: one counter per layer, all rows at the same
    length); ``starts[b]`` shifts row b's position ids (a left-padded batch): the forward hands position_ids[:, 0] to the
    RoPE kernel, one offset per row (llama.py:347-352)."""
    from duo_attn.patch.llama import llama_duo_attention_forward_one_way_reordered_static as fwd
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    D, L = 128, len(counts)
    heads = [[1.0] * nf + [0.0] * (Hkv - nf) for nf in counts]
    model = types.SimpleNamespace(
        config=types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv,
                                     hidden_size=Hq * D),
        parameters=lambda: iter([torch.zeros(1, dtype=torch.bfloat16)]),
    )
    total = sum(chunks) + decode_steps + 2
    starts = [0] * batch if starts is None else list(starts)
    cache = DuoAttentionStaticKVCache(model, heads, batch, total, sink, recent)
    attn = fake_attention(Hq, Hkv, D, theta, factor)
    g = torch.Generator().manual_seed(seed)
    out = {"counts": np.array(counts), "dims": np.array([Hq, Hkv, D, sink, recent]),
           "rope": np.array([theta, 1.0 if factor is None else factor], dtype=np.float64),
           "steps": np.array(list(chunks) + [1] * decode_steps), "n_prefill": np.array(len(chunks))}
    if batch > 1:
        out["starts"] = np.array(starts)
    pos = 0
    for si, S in enumerate(list(chunks) + [1] * decode_steps):
        position_ids = torch.stack([torch.arange(pos + s0, pos + s0 + S) for s0 in starts])
        for l in range(L):
            h = torch.randn(batch, S, Hq * D, generator=g).to(torch.bfloat16)
            out[f"h_{si}_{l}"] = bits(h)   # before the call: q aliases h and is rotated in place
            o, _ = fwd(attn, h, position_ids=position_ids, kv_cache=cache, layer_idx=l)
            out[f"o_{si}_{l}"] = bits(o)
        if si >= len(chunks):
            cache.evict_last(1)
        else:
            pos += S
    for l in range(L):
        n, m = cache.kv_seq_len_list[l], cache.streaming_kv_seq_len_list[l]
        out[f"len_{l}"] = np.array([n, m])
        out[f"fullk_{l}"] = bits(cache.full_key_states_list[l][:, :n])
        out[f"fullv_{l}"] = bits(cache.full_value_states_list[l][:, :n])
        out[f"strk_{l}"] = bits(cache.streaming_key_states_list[l][:, :m])
        out[f"strv_{l}"] = bits(cache.streaming_value_states_list[l][:, :m])
    np.savez_compressed(os.path.join(HERE, name), **out)


def golden_tuple(name, nf, Hq, Hkv, chunks, sink, recent, theta, seed):
    """The tuple-cache forward (llama.py:146-306) with HF's rotary (current transformers
    apply_rotary_pos_emb, bf16 cos/sin computed the HF way)."""
    from duo_attn.patch.llama import llama_duo_attention_forward_one_way_reordered as fwd

    D = 128
    attn = fake_attention(Hq, Hkv, D, theta, None)
    inv_freq = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))

    def rotary_emb(x, position_ids):
        freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos().to(x.dtype), emb.sin().to(x.dtype)

    attn.rotary_emb = rotary_emb
    attn.sink_size, attn.recent_size = sink, recent
    attn.register_buffer("full_attention_heads", torch.tensor([1.0] * nf + [0.0] * (Hkv - nf)))
    g = torch.Generator().manual_seed(seed)
    out = {"dims": np.array([Hq, Hkv, D, sink, recent, nf]), "theta": np.array(theta), "steps": np.array(chunks)}
    past, pos = None, 0
    for si, S in enumerate(chunks):
        h = torch.randn(1, S, Hq * D, generator=g).to(torch.bfloat16)
        out[f"h_{si}"] = bits(h)
        o, _, past = fwd(attn, h, position_ids=torch.arange(pos, pos + S)[None], past_key_value=past, use_cache=True)
        out[f"o_{si}"] = bits(o)
        pos += S
    out["past_full"] = bits(past[0])
    out["past_stream"] = bits(past[1])
    np.savez_compressed(os.path.join(HERE, name), **out)


def golden_layer(name, nf, Hq, Hkv, inter, chunks, decode_steps, sink, recent, theta, seed):
    """A whole DECODER LAYER through the reference's own code: ``duo_attn_static_kv_cache_llama_decoder_layer_forward``
    (static_kv_cache.py:507-546) around its static attention forward (llama.py:309-434), with real ``nn.Linear``
    projections, HF's ``LlamaMLP`` and ``LlamaRMSNorm`` modules carrying the reference's ``flashinfer_rmsnorm_forward``
    (flashinfer_utils.py:9-16; the arithmetic stub above), its real DuoAttentionStaticKVCache, bf16 on the CPU.  Pins the
    module SEQUENCE either side of the attention op — norm, q/k/v_proj, o_proj, residual, norm, SwiGLU MLP, residual —
    which this repository runs as fused launches at q_len == 1 (csrc/duo_linear.hip, DESIGN row (g))."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaMLP, LlamaRMSNorm

    from duo_attn.patch.flashinfer_utils import flashinfer_rmsnorm_forward
    from duo_attn.patch.llama import llama_duo_attention_forward_one_way_reordered_static as attn_fwd
    from duo_attn.patch.static_kv_cache import (DuoAttentionStaticKVCache,
                                                duo_attn_static_kv_cache_llama_decoder_layer_forward as layer_fwd)

    D, H = 128, Hq * 128
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    heads = [[1.0] * nf + [0.0] * (Hkv - nf)]
    model = types.SimpleNamespace(
        config=types.SimpleNamespace(num_hidden_layers=1, num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=H),
        parameters=lambda: iter([torch.zeros(1, dtype=torch.bfloat16)]),
    )
    total = sum(chunks) + decode_steps + 2
    cache = DuoAttentionStaticKVCache(model, heads, 1, total, sink, recent)
    attn = fake_attention(Hq, Hkv, D, theta, None)
    rw = lambda o, i: (torch.randn(o, i, generator=g) * i ** -0.5).to(torch.bfloat16)
    for nm, o, i in (("q_proj", H, H), ("k_proj", Hkv * D, H), ("v_proj", Hkv * D, H), ("o_proj", H, H)):
        lin = torch.nn.Linear(i, o, bias=False).to(torch.bfloat16)
        lin.weight.data.copy_(rw(o, i))
        setattr(attn, nm, lin)
    attn.forward = types.MethodType(attn_fwd, attn)
    layer = torch.nn.Module()
    layer.self_attn = attn
    layer.mlp = LlamaMLP(LlamaConfig(hidden_size=H, intermediate_size=inter)).to(torch.bfloat16)
    for lin in (layer.mlp.gate_proj, layer.mlp.up_proj, layer.mlp.down_proj):
        lin.weight.data.copy_(rw(*lin.weight.shape))
    for nm in ("input_layernorm", "post_attention_layernorm"):
        ln = LlamaRMSNorm(H, eps=1e-5).to(torch.bfloat16)
        ln.weight.data.copy_((torch.rand(H, generator=g) + 0.5).to(torch.bfloat16))
        ln.forward = types.MethodType(flashinfer_rmsnorm_forward, ln)
        setattr(layer, nm, ln)
    out = {"dims": np.array([Hq, Hkv, D, sink, recent, inter, nf]), "rope": np.array([theta, 1.0], dtype=np.float64),
           "eps": np.array(1e-5), "steps": np.array(list(chunks) + [1] * decode_steps), "n_prefill": np.array(len(chunks))}
    for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
        out[f"w_{nm}"] = bits(getattr(attn, nm).weight.data)
    for nm in ("gate_proj", "up_proj", "down_proj"):
        out[f"w_{nm}"] = bits(getattr(layer.mlp, nm).weight.data)
    out["w_input_layernorm"] = bits(layer.input_layernorm.weight.data)
    out["w_post_attention_layernorm"] = bits(layer.post_attention_layernorm.weight.data)
    pos = 0
    with torch.no_grad():
        for si, S in enumerate(list(chunks) + [1] * decode_steps):
            position_ids = torch.arange(pos, pos + S)[None]
            h = torch.randn(1, S, H, generator=g).to(torch.bfloat16)
            out[f"h_{si}"] = bits(h)
            o = layer_fwd(layer, h.clone(), position_ids=position_ids, kv_cache=cache, layer_idx=0)[0]
            out[f"o_{si}"] = bits(o)
            pos += S              # (no evict_last: the decode steps advance the cache like a generation loop)
    n, m = cache.kv_seq_len_list[0], cache.streaming_kv_seq_len_list[0]
    out["len"] = np.array([n, m])
    out["fullv"] = bits(cache.full_value_states_list[0][:, :n])
    out["strv"] = bits(cache.streaming_value_states_list[0][:, :m])
    np.savez_compressed(os.path.join(HERE, name), **out)


def golden_host(name):
    """Host-side pieces: pattern sparsification on the shipped TSVs and the weight reordering."""
    from duo_attn.patch.utils import reorder_full_attn_heads, reorder_linear_weights
    from duo_attn.utils import load_attn_pattern, seed_everything, sparsify_attention_heads

    out = {}
    pat_root = os.path.join(REF, "attn_patterns")
    for model in sorted(os.listdir(pat_root)):
        run = sorted(os.listdir(os.path.join(pat_root, model)))[0]
        for sparsity in (0.5, 0.75):
            seed_everything(42)
            heads, sink, recent = load_attn_pattern(os.path.join(pat_root, model, run))
            raw = heads.copy()
            state = np.random.get_state()
            mask, true_sparsity = sparsify_attention_heads(heads, None, sparsity)
            key = f"{model}@{sparsity}"
            out[f"raw|{model}"] = raw
            out[f"mask|{key}"] = mask
            out[f"meta|{key}"] = np.array([sink, recent, true_sparsity])
            np.random.set_state(state)
            out[f"noise|{key}"] = np.random.uniform(0, 1e-6, raw.shape)
    g = torch.Generator().manual_seed(7)
    heads = torch.tensor([0.0, 1.0, 0.0, 1.0])
    for chan, rep in (("out", 6), ("in", 6), ("out", 2)):
        lin = torch.nn.Linear(24 if chan == "in" else 5, 24 if chan == "out" and rep == 6 else (8 if chan == "out" else 5),
                              bias=(chan == "out"))
        with torch.no_grad():
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=g))
            if lin.bias is not None:
                lin.bias.copy_(torch.randn(lin.bias.shape, generator=g))
        out[f"lin_w_before|{chan}{rep}"] = lin.weight.detach().numpy().copy()
        if lin.bias is not None:
            out[f"lin_b_before|{chan}{rep}"] = lin.bias.detach().numpy().copy()
        lin = reorder_linear_weights(lin, heads.clone(), rep, chan)
        out[f"lin_w_after|{chan}{rep}"] = lin.weight.detach().numpy().copy()
        if lin.bias is not None:
            out[f"lin_b_after|{chan}{rep}"] = lin.bias.detach().numpy().copy()
    out["reordered_heads"] = reorder_full_attn_heads(torch.tensor([0.2, 0.9, 0.4, 0.7, 1.0])).numpy()
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    install_shims()
    torch.set_num_threads(8)
    golden_host("host.npz")
    # ragged split incl. nf=0 and nf=Hkv; pool saturates (sink 4 + recent 12) during the second chunk
    golden_static("static_a.npz", counts=[1, 0, 4, 2], Hq=8, Hkv=4, chunks=(24, 17, 9), decode_steps=3,
                  sink=4, recent=12, theta=10000.0, factor=None, seed=1)
    # MHA with linear rope factor 8 (Llama-2-7B-32K style), shipped sink/recent sizes
    golden_static("static_b.npz", counts=[3, 1], Hq=4, Hkv=4, chunks=(300, 200), decode_steps=2,
                  sink=128, recent=256, theta=10000.0, factor=8.0, seed=2)
    # the reference's batch dimension, rows starting at different positions (position_ids[:, 0] per row)
    golden_static("static_c.npz", counts=[2, 0, 3], Hq=8, Hkv=4, chunks=(40, 21), decode_steps=3,
                  sink=4, recent=12, theta=500000.0, factor=None, seed=4, batch=2, starts=[0, 7])
    golden_tuple("tuple_a.npz", nf=1, Hq=8, Hkv=4, chunks=(20, 9, 1, 1), sink=4, recent=8, theta=10000.0, seed=3)
    # a whole decoder layer (norms, projections, MLP, residual adds around the static attention forward)
    golden_layer("layer_a.npz", nf=1, Hq=4, Hkv=2, inter=256, chunks=(29, 11), decode_steps=5, sink=4, recent=12,
                 theta=10000.0, seed=6)
    print("golden vectors written to", HERE)
