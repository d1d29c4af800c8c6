"""Family-agnostic DuoAttention forwards and enablers (Llama == Mistral).

The reference keeps two 693-line copies (``duo_attn/patch/llama.py`` and
``mistral.py``) that differ only by ``s/llama/mistral/``; here one
implementation is parameterised by the HF module classes and re-exported under
both names by ``llama.py`` / ``mistral.py``.

Every attention call goes through ``backend.get_backend()`` — the HIP C-ABI
library.  The projections (``nn.Linear`` -> hipBLASLt) are HF's own, except at
q_len == 1 on the static path, where they are matrix-vector products streamed by
``duo_token_linear_bf16`` (``duo_decode_layer_fused`` below).

Written against transformers 5.x: the attention module no longer has
``num_heads`` / ``rotary_emb`` / ``rope_theta`` (reference llama.py:65,68,87,
132,351) — shapes come from ``module.config`` and the tuple path's cos/sin are
computed once per forward by ``model.rotary_emb``.
"""
from __future__ import annotations

import os
import types
from typing import Optional, Tuple

import torch

from ..backend import get_backend
from .flashinfer_utils import apply_rope_inplace, enable_flashinfer_rmsnorm
from .static_kv_cache import DuoAttentionStaticKVCache, enable_duo_attention_static_kv_cache
from .tuple_kv_cache import enable_tuple_kv_cache_for_model, hf_apply_rotary_pos_emb, tuple_rotary
from .utils import reorder_full_attn_heads, reorder_linear_weights


def _dims(module):
    cfg = module.config
    num_heads = cfg.num_attention_heads
    num_kv = cfg.num_key_value_heads
    head_dim = getattr(module, "head_dim", cfg.hidden_size // num_heads)
    return num_heads, num_kv, head_dim, num_heads // num_kv


def rope_scale_and_theta(config) -> Tuple[float, float]:
    """rope_scale / rope_theta exactly as the static path reads them (reference llama.py:347-352):
    ``rope_scaling["factor"]`` whatever the rope_type (linear scaling — a quirk that is part of the
    contract), ``rope_theta`` from the config.  transformers 5.x folds both into ``rope_parameters``."""
    rp = getattr(config, "rope_parameters", None) or {}
    theta = rp.get("rope_theta", None)
    if theta is None:
        theta = getattr(config, "rope_theta", 10000.0)
    scaling = getattr(config, "rope_scaling", None)
    if scaling is None:
        scaling = rp
    scale = 1.0
    if scaling:
        f = scaling.get("factor", 1.0)
        scale = 1.0 if f is None else float(f)
    return scale, float(theta)


def first_positions(position_ids):
    """``position_ids[:, 0]`` as host values for the RoPE launch (one read-back): an int when every batch row
    starts at the same position (every reference harness), else the per-row list."""
    first = position_ids.reshape(position_ids.shape[0], -1)[:, 0].tolist()
    return int(first[0]) if all(f == first[0] for f in first) else [int(f) for f in first]


def duo_static_attention_core(query_states, key_states, value_states, kv_cache, layer_idx, pos0,
                              rope_scale, rope_theta):
    """The hot path proper: everything between the q/k/v projections and o_proj in reference
    llama.py:332-425 — RoPE in place, head split, full-pool append, split-head attention,
    streaming-pool update.  q [B,S,Hq,D], k/v [B,S,Hkv,D] bf16; returns [B,S,Hq,D].
    ``pos0``: position of the first new row — an int, or one int per batch row (the reference hands
    ``position_ids[:, 0]`` to the RoPE kernel, llama.py:350-352); None = the cache length."""
    bsz, q_len, num_heads, head_dim = query_states.shape
    num_kv = key_states.shape[2]
    groups = num_heads // num_kv

    past = kv_cache.kv_seq_len  # last layer's counter, as the reference reads it (:105-107)
    kv_seq_len = q_len + past
    if pos0 is None:
        pos0 = past
    be = get_backend()
    if q_len == 1 and past > 0 and hasattr(be, "decode_layer"):
        return _decode_step_fused(be, query_states, key_states, value_states, kv_cache, layer_idx, pos0,
                                  rope_scale, rope_theta)
    apply_rope_inplace(query_states, key_states, pos0, rope_scale, rope_theta)

    fk, fv, sk, sv = kv_cache.split_kv(layer_idx, key_states, value_states)
    past_l = kv_cache.kv_seq_len_list[layer_idx]
    kv_cache.put_full_kv(layer_idx, fk, fv)

    attn_output = torch.empty_like(query_states)
    scale = head_dim ** -0.5
    batched = bsz > 1 and hasattr(be, "attention_batched")      # one launch for all batch rows (the C ABI's batched forms)
    if q_len == kv_seq_len:
        # initial pre-filling: every head is causal over the chunk (:364-372)
        if batched:
            be.attention_batched(query_states, attn_output, groups, (num_kv, 0, None, (key_states, value_states)), None, scale)
        for b in range(0 if batched else bsz):
            be.attention(query_states[b], attn_output[b], groups,
                         (num_kv, 0, None, (key_states[b], value_states[b])), None, scale)
    else:
        # decoding or continued filling (:374-421): retrieval heads over the full pool,
        # streaming heads over [pool ++ new rows]; the concat is two kernel segments.
        nf = kv_cache.num_full_kv_head_list[layer_idx]
        ns = num_kv - nf
        pk, pv = kv_cache.full_key_states_list[layer_idx], kv_cache.full_value_states_list[layer_idx]
        ck, cv = kv_cache.get_streaming_kv(layer_idx)
        if batched:
            full = (nf, 0, (pk[:, :past_l], pv[:, :past_l]),
                    (pk[:, past_l:past_l + q_len], pv[:, past_l:past_l + q_len])) if nf > 0 else None
            stream = (ns, nf * groups, (ck, cv), (sk, sv)) if ns > 0 else None
            be.attention_batched(query_states, attn_output, groups, full, stream, scale)
        for b in range(0 if batched else bsz):
            full = (nf, 0, (pk[b, :past_l], pv[b, :past_l]),
                    (pk[b, past_l:past_l + q_len], pv[b, past_l:past_l + q_len])) if nf > 0 else None
            stream = (ns, nf * groups, (ck[b], cv[b]), (sk[b], sv[b])) if ns > 0 else None
            be.attention(query_states[b], attn_output[b], groups, full, stream, scale)

    kv_cache.update_streaming_kv(layer_idx, sk, sv)
    return attn_output

def duo_static_attention_row_block(query_states, key_states, value_states, kv_cache, layer_idx, r0, chunk_len,
                                   rope_scale, rope_theta):
    """Rows ``[r0, r0 + n)`` of a prefill chunk of ``chunk_len`` rows — the same mathematics as
    ``duo_static_attention_core`` on the whole chunk, issued block by block in row order.

    Why: in a layer pipeline a stage can start on the first rows of a chunk as soon as the previous stage
    has produced them (causality: row r only needs rows <= r of its own chunk), so the pipeline fills in
    row-block steps instead of whole-chunk steps.  Semantics are the chunk's, not a smaller chunk's: a
    streaming head still sees the pool as it was at the START of the chunk plus every earlier row of the
    chunk (reference llama.py:374-421), which is why the chunk's streaming K/V rows are kept in a staging
    buffer until the last block, when the pool update runs once on the whole chunk (:423-425).
    q [B,n,Hq,D], k/v [B,n,Hkv,D] (un-rotated); blocks of one chunk must come in order, ``r0 == 0`` first."""
    bsz, n, num_heads, head_dim = query_states.shape
    num_kv = key_states.shape[2]
    groups = num_heads // num_kv
    nf = kv_cache.num_full_kv_head_list[layer_idx]
    ns = num_kv - nf
    r1 = r0 + n
    if r1 > chunk_len:
        raise ValueError(f"row block [{r0}, {r1}) exceeds the chunk of {chunk_len} rows")
    st = kv_cache.begin_chunk(layer_idx, chunk_len) if r0 == 0 else kv_cache.chunk_state(layer_idx)
    if st["next_row"] != r0 or st["chunk_len"] != chunk_len:
        raise ValueError(f"row blocks out of order: expected row {st['next_row']} of a {st['chunk_len']}-row chunk")
    past_l = st["past"]
    be = get_backend()
    apply_rope_inplace(query_states, key_states, past_l + r0, rope_scale, rope_theta)
    fk, fv, sk, sv = kv_cache.split_kv(layer_idx, key_states, value_states)
    kv_cache.put_full_kv(layer_idx, fk, fv)                 # lands at rows past_l + r0 .. of the full pool
    stage_k, stage_v = st["stage_k"], st["stage_v"]
    if ns > 0:
        stage_k[:, r0:r1].copy_(sk)
        stage_v[:, r0:r1].copy_(sv)
    attn_output = torch.empty_like(query_states)
    scale = head_dim ** -0.5
    pk, pv = kv_cache.full_key_states_list[layer_idx], kv_cache.full_value_states_list[layer_idx]
    ck, cv = kv_cache.get_streaming_kv(layer_idx)           # unchanged until the chunk's last block
    batched = bsz > 1 and hasattr(be, "attention_batched")      # one launch for all batch rows, as in the whole-chunk path

    def classes(sel):
        """the two head classes of batch row(s) ``sel`` (an index, or slice(None) for the batched launch)"""
        if past_l == 0:     # first chunk: every head causal over the chunk's own rows (:364-372)
            full = (nf, 0, None, (pk[sel, :r1], pv[sel, :r1])) if nf > 0 else None
            stream = (ns, nf * groups, None, (stage_k[sel, :r1], stage_v[sel, :r1])) if ns > 0 else None
        else:
            full = (nf, 0, (pk[sel, :past_l], pv[sel, :past_l]),
                    (pk[sel, past_l:past_l + r1], pv[sel, past_l:past_l + r1])) if nf > 0 else None
            stream = (ns, nf * groups, (ck[sel], cv[sel]), (stage_k[sel, :r1], stage_v[sel, :r1])) if ns > 0 else None
        return full, stream

    if batched:
        be.attention_batched(query_states, attn_output, groups, *classes(slice(None)), scale)
    for b in range(0 if batched else bsz):
        be.attention(query_states[b], attn_output[b], groups, *classes(b), scale)
    st["next_row"] = r1
    if r1 == chunk_len:
        kv_cache.update_streaming_kv(layer_idx, stage_k[:, :chunk_len], stage_v[:, :chunk_len])
        kv_cache.end_chunk(layer_idx)
    return attn_output


def _decode_step_fused(be, query_states, key_states, value_states, kv_cache, layer_idx, pos0, rope_scale,
                       rope_theta):
    """q_len == 1 after a prefill: the same steps as the general path below (RoPE, put_full_kv, the two
    head-class attentions, the streaming update) issued as ONE backend call = two kernel launches,
    with the cache counters updated exactly as put_full_kv / compress_and_replace_streaming_kv do."""
    bsz, _, num_heads, head_dim = query_states.shape
    nf = kv_cache.num_full_kv_head_list[layer_idx]
    cur = kv_cache.kv_seq_len_list[layer_idx]
    if nf > 0 and cur + 1 > kv_cache.max_size:
        raise ValueError(
            f"Trying to put 1 KVs into a cache with max size {kv_cache.max_size}, current size: {cur}."
        )
    str_len = kv_cache.streaming_kv_seq_len_list[layer_idx]
    pk, pv = kv_cache.full_key_states_list[layer_idx], kv_cache.full_value_states_list[layer_idx]
    sk, sv = kv_cache.streaming_key_states_list[layer_idx], kv_cache.streaming_value_states_list[layer_idx]
    attn_output = torch.empty_like(query_states)
    W = kv_cache.sink_size + kv_cache.recent_size
    new_len = min(str_len + 1, W)
    pos_rows = list(pos0) if isinstance(pos0, (list, tuple)) else [pos0] * bsz
    if getattr(kv_cache, "use_device_state", False):
        # graph-capturable form: the kernels read full_len / str_len / pos from the layer's device state;
        # the host values passed here only size the split-KV grid
        if bsz == 1:
            be.decode_layer_dev(query_states[0, 0], key_states[0, 0], value_states[0, 0], attn_output[0, 0], nf,
                                pk[0], pv[0], cur, sk[0], sv[0], str_len, kv_cache.sink_size, kv_cache.recent_size,
                                pos_rows[0], rope_scale, rope_theta, head_dim ** -0.5, kv_cache.device_state[layer_idx])
        else:
            # all batch rows share the layer's counters (reference static_kv_cache.py:44-45); a row that started at another
            # position keeps its fixed offset from row 0 (pos_rows[b] - pos_rows[0]) on top of the device-side position
            be.decode_layer_batched_dev(query_states[:, 0], key_states[:, 0], value_states[:, 0], attn_output[:, 0], nf,
                                        pk, pv, cur, sk, sv, str_len, kv_cache.sink_size, kv_cache.recent_size,
                                        pos_rows, rope_scale, rope_theta, head_dim ** -0.5,
                                        kv_cache.device_state[layer_idx])
    elif bsz > 1 and hasattr(be, "decode_layer_batched"):
        # every batch row in ONE launch pair (grid-level batch rows; rows at different positions fall back to a launch
        # pair per row inside the library)
        new_len = be.decode_layer_batched(query_states[:, 0], key_states[:, 0], value_states[:, 0], attn_output[:, 0], nf,
                                          pk, pv, cur, sk, sv, str_len, kv_cache.sink_size, kv_cache.recent_size,
                                          pos_rows, rope_scale, rope_theta, head_dim ** -0.5)
    else:
        for b in range(bsz):
            new_len = be.decode_layer(query_states[b, 0], key_states[b, 0], value_states[b, 0], attn_output[b, 0], nf,
                                      pk[b], pv[b], cur, sk[b], sv[b], str_len, kv_cache.sink_size,
                                      kv_cache.recent_size, pos_rows[b], rope_scale, rope_theta, head_dim ** -0.5)
    kv_cache.kv_seq_len_list[layer_idx] = cur + 1
    kv_cache.streaming_kv_seq_len_list[layer_idx] = new_len
    return attn_output


# =============================================================================
# decode step of a decoder layer with the token-row linears fused around the attention op
# =============================================================================
_FUSED_DECODE_LAYER = os.environ.get("DUO_FUSED_DECODE_LAYER", "1") != "0"     # (0: module by module, for A/B)


def modules_hooked(mods) -> bool:
    """whether a forward (pre-)hook is registered on any of ``mods`` (or globally): the fused forms go around these modules'
    ``__call__`` — with a hook on one of them the step takes the module sequence, so the hook fires as it does in the reference"""
    import torch.nn.modules.module as nnm

    if nnm._global_forward_hooks or nnm._global_forward_pre_hooks:
        return True
    return any(getattr(m, "_forward_hooks", None) or getattr(m, "_forward_pre_hooks", None) for m in mods)


def _row_parallel(m):
    """a tensor-parallel shard's o_proj / down_proj (duo_attn/tp.py: RowParallelLinear): the local slice + the group"""
    return type(m).__name__ == "RowParallelLinear" and hasattr(m, "inner") and hasattr(m, "group")


def _streamable_linear(m) -> bool:
    if _row_parallel(m):
        m = m.inner
    w = getattr(m, "weight", None)
    return (type(m) is torch.nn.Linear and w is not None and w.dtype == torch.bfloat16 and w.is_cuda
            and w.stride(1) == 1 and w.stride(0) % 8 == 0 and w.data_ptr() % 16 == 0
            and (m.bias is None or (m.bias.dtype == torch.bfloat16 and m.bias.is_contiguous())))


def _layer_modules(layer):
    attn, mlp = layer.self_attn, getattr(layer, "mlp", None)
    return [getattr(attn, n, None) for n in ("q_proj", "k_proj", "v_proj", "o_proj")] + \
           [getattr(mlp, n, None) for n in ("gate_proj", "up_proj", "down_proj")]


def _norm_form(norm):
    """which arithmetic an RMSNorm module's forward performs: "flashinfer" (one rounding: the static path's patched forward,
    flashinfer_utils.rmsnorm_forward), "hf" (HuggingFace's own LlamaRMSNorm / MistralRMSNorm forward, two roundings — what
    the tuple path runs, the reference's enable_duo_attention_eval leaves the norms alone), or None (anything else)"""
    from .flashinfer_utils import rmsnorm_forward

    w = getattr(norm, "weight", None)
    if not hasattr(norm, "variance_epsilon") or w is None or w.dtype != torch.bfloat16 or not w.is_contiguous():
        return None
    fwd = getattr(norm.forward, "__func__", None)
    if fwd is rmsnorm_forward:
        return "flashinfer"
    if type(norm).__name__ in ("LlamaRMSNorm", "MistralRMSNorm") and fwd is type(norm).forward:
        return "hf"
    return None


class _FusedRefs:
    """What the fused decode layer reads from a decoder layer, looked up ONCE: torch.nn.Module attribute access goes through
    ``__getattr__`` (a dict walk per access, ~70 of them per layer and token otherwise).  Parameters are held as objects —
    their ``.data`` may be swapped (weight reorder), the launch reads ``data_ptr()`` at call time."""

    __slots__ = ("key", "ok", "attn", "qkv", "o", "gu", "down", "n_w", "n_eps", "n_hf", "p_w", "p_eps", "p_hf", "dims", "inter",
                 "in_feats", "fits", "bypassed")



def _out_linear(be, proj, x, x2, residual):
    """o_proj / down_proj of the fused decode layer: ``proj(x [* silu-partner x2]) + residual`` in one launch, or — row
    parallel (a tensor-parallel shard, duo_attn.tp) — the local product, the all-reduce over the TP group, then the add"""
    if not _row_parallel(proj):
        return be.token_linear(x, [(proj.weight, proj.bias)], x2=x2, residual=residual)
    from ..tp import all_reduce_sum

    y = be.token_linear(x, [(proj.inner.weight, proj.inner.bias)], x2=x2)
    return residual + all_reduce_sum(y, proj.group)


def _refs_key(layer, want_fwd):
    attn = layer.self_attn
    return (attn.q_proj.weight.data_ptr(), layer.mlp.down_proj.weight.data_ptr(),
            getattr(attn.forward, "__func__", None) is want_fwd,
            id(getattr(layer.input_layernorm.forward, "__func__", None)),
            id(getattr(layer.post_attention_layernorm.forward, "__func__", None)))


def _fused_refs(layer, want_fwd=None, tag="static"):
    """the layer's ``_FusedRefs`` for the attention forward ``want_fwd`` (default: this module's static forward), cached on
    the layer and keyed by the storage of its first and last projection and the identity of the forwards involved — the
    enablers, a weight reload (``.to()``, ``load_state_dict`` into new storage) or a re-shard re-evaluate it.  ``.ok`` says
    whether the layer's modules allow the fused form at all: attention carrying THIS module's forward (i.e. it went through
    the matching enabler, weights reordered retrieval-heads-first), bf16 ``nn.Linear`` projections on the GPU (or a TP
    shard's row-parallel o_proj / down_proj), SiLU-gated MLP, RMSNorm modules in a known form."""
    if want_fwd is None:
        want_fwd = duo_attention_forward_one_way_reordered_static
    cache = layer.__dict__.get("_duo_fused_refs")
    if cache is None:
        cache = layer.__dict__["_duo_fused_refs"] = {}
    hit = cache.get(tag)
    try:
        key = _refs_key(layer, want_fwd)
    except AttributeError:          # not a Llama-style decoder layer
        key = None
    if hit is not None and hit.key == key and key is not None:
        return hit
    r = _FusedRefs()
    r.key, r.ok = key, False
    cache[tag] = r
    if key is None:
        return r
    mods = _layer_modules(layer)
    mlp, attn = layer.mlp, layer.self_attn
    n_ln, p_ln = layer.input_layernorm, layer.post_attention_layernorm
    if any(m is None or getattr(m, "weight", None) is None for m in mods) or not hasattr(n_ln, "variance_epsilon") \
            or not hasattr(p_ln, "variance_epsilon"):
        return r
    # (the references are filled in whenever the layer has the shape of a Llama / Mistral decoder layer — the CPU test-suite
    #  drives the fused form's host logic through them with the oracle as backend; `ok` is the product's own gate)
    forms = (_norm_form(n_ln), _norm_form(p_ln))
    lin = lambda m: (m.inner if _row_parallel(m) else m)
    wb = lambda m: (lin(m).weight, lin(m).bias)
    r.attn = attn
    r.bypassed = [attn, mlp, n_ln, p_ln] + [x for m in mods for x in ((m, m.inner) if _row_parallel(m) else (m,))]   # modules_hooked
    r.qkv = [wb(attn.q_proj), wb(attn.k_proj), wb(attn.v_proj)]
    r.gu = [wb(mlp.gate_proj), wb(mlp.up_proj)]
    r.o, r.down = attn.o_proj, mlp.down_proj            # (modules: _out_linear decides plain / row-parallel)
    r.n_w, r.n_eps, r.n_hf = n_ln.weight, n_ln.variance_epsilon, forms[0] == "hf"
    r.p_w, r.p_eps, r.p_hf = p_ln.weight, p_ln.variance_epsilon, forms[1] == "hf"
    r.dims = _dims(attn)
    r.inter = lin(mlp.gate_proj).out_features
    r.in_feats = sorted({lin(m).in_features for m in mods})
    r.fits = {}
    r.ok = (key[2] and type(getattr(mlp, "act_fn", None)).__name__ in ("SiLUActivation", "SiLU")
            and all(_streamable_linear(m) for m in mods) and None not in forms)
    return r


def _layer_static_verdict(layer, want_fwd=None, key_tag="static") -> bool:
    return _fused_refs(layer, want_fwd, key_tag).ok


def _rows_fit(be, refs, rows) -> bool:
    ok = refs.fits.get(rows)
    if ok is None:
        ok = refs.fits[rows] = all(be.token_linear_fits(rows, n) for n in refs.in_feats)
    return ok


def fused_decode_layer_ok(layer, hidden_states, kv_cache, layer_idx) -> bool:
    """Whether this decoder layer's decode step can run as the fused form below: one token per batch row after a
    prefill, bf16 on the GPU, ``nn.Linear`` projections (or a tensor-parallel shard's row-parallel o_proj / down_proj),
    SiLU-gated MLP, RMSNorm modules with a known forward, a backend that has the kernel."""
    if not _FUSED_DECODE_LAYER or hidden_states.dim() != 3 or hidden_states.shape[1] != 1:
        return False
    be = get_backend()
    if not hasattr(be, "token_linear") or not hidden_states.is_cuda or hidden_states.dtype != torch.bfloat16:
        return False
    if not isinstance(kv_cache, DuoAttentionStaticKVCache) or kv_cache.kv_seq_len_list[layer_idx] <= 0:
        return False
    if hidden_states.stride(2) != 1 or hidden_states.stride(0) % 8 or hidden_states.data_ptr() % 16:
        return False                       # (the kernel's 16-byte row loads)
    refs = _fused_refs(layer)
    return refs.ok and _rows_fit(be, refs, hidden_states.shape[0]) and not modules_hooked(refs.bypassed)


def _norm_kw(hf: bool):
    """``norm_hf=True`` for a norm module that runs HuggingFace's own two-rounding forward (``_norm_form``)"""
    return {"norm_hf": True} if hf else {}


def duo_decode_layer_fused(layer, hidden_states, kv_cache, layer_idx, pos0=None, position_ids=None):
    """The reference's decoder layer at q_len == 1 (static_kv_cache.py:482-537 around llama.py:309-434) in six launches:

        q|k|v = [Wq; Wk; Wv] . rmsnorm(h)            input_layernorm folded into the projection's prologue
        attn  = duo_static_attention_core(...)       RoPE + append + split-head attention + streaming update (2 launches)
        h1    = Wo . attn + h                        first residual add folded into o_proj's epilogue
        g|u   = [Wg; Wu] . rmsnorm(h1)               post_attention_layernorm folded in
        h2    = Wd . (silu(g) * u) + h1              SiLU*mul in the prologue, second residual add in the epilogue

    Every value the modules materialise as a bf16 tensor is rounded to bf16 at the same point (csrc/duo_linear.hip);
    the linears differ from the library GEMM only by the summation order of their fp32 dot products.

    Tensor-parallel shard (``duo_attn.tp.shard_model_for_tp``): o_proj / down_proj are row-parallel — the local product
    is all-reduced over the TP group BEFORE the residual add (reference ``tensor_parallel`` semantics, utils.py:206-227),
    so their residual epilogue becomes a separate add behind the all-reduce; the other two launches are unchanged."""
    be = get_backend()
    r = _fused_refs(layer)
    bsz, _, hidden = hidden_states.shape
    num_heads, num_kv, head_dim, _ = r.dims
    x = hidden_states.reshape(bsz, hidden)
    qkv = be.token_linear(x, r.qkv, norm=(r.n_w, r.n_eps), **_norm_kw(r.n_hf))
    nq, nk = num_heads * head_dim, num_kv * head_dim
    q = qkv[:, :nq].view(bsz, 1, num_heads, head_dim)
    k = qkv[:, nq:nq + nk].view(bsz, 1, num_kv, head_dim)
    v = qkv[:, nq + nk:].view(bsz, 1, num_kv, head_dim)
    rope = r.attn.__dict__.get("_duo_rope")
    if rope is None:
        rope = r.attn.__dict__["_duo_rope"] = rope_scale_and_theta(r.attn.config)
    if pos0 is None and position_ids is not None:
        pos0 = first_positions(position_ids)
    ao = duo_static_attention_core(q, k, v, kv_cache, layer_idx, pos0, rope[0], rope[1])
    h1 = _out_linear(be, r.o, ao.reshape(bsz, nq), None, x)
    gu = be.token_linear(h1, r.gu, norm=(r.p_w, r.p_eps), **_norm_kw(r.p_hf))
    h2 = _out_linear(be, r.down, gu[:, :r.inter], gu[:, r.inter:], h1)
    return h2.view(bsz, 1, hidden)


# =============================================================================
# decode step of a decoder layer on the TUPLE cache (enable_duo_attention_eval), fused the same way
# =============================================================================
def tuple_fused_decode_ok(layer, hidden_states, past_key_value, position_embeddings, use_cache) -> bool:
    """Whether this decoder layer's decode step on the tuple cache can run as ``duo_tuple_decode_layer_fused``: one token,
    one batch row, a non-empty past in the tuple format, bf16 on the GPU, ``nn.Linear`` projections, SiLU-gated MLP, RMSNorm
    modules in a known form, HF rotary cos / sin for this position, a backend that has the kernels."""
    if not _FUSED_DECODE_LAYER or not use_cache or hidden_states.dim() != 3 or hidden_states.shape[:2] != (1, 1):
        return False
    if past_key_value is None or position_embeddings is None or len(past_key_value) != 2:
        return False
    be = get_backend()
    if not (hasattr(be, "token_linear") and hasattr(be, "tuple_decode_prep")):
        return False
    if not hidden_states.is_cuda or hidden_states.dtype != torch.bfloat16:
        return False
    if hidden_states.stride(2) != 1 or hidden_states.data_ptr() % 16:
        return False
    if torch.is_grad_enabled() and (hidden_states.requires_grad or layer.self_attn.q_proj.weight.requires_grad):
        return False        # the ctypes kernels build no autograd graph: the module path does (as tuple_rotary / _hf_norm)
    refs = _fused_refs(layer, duo_attention_forward_one_way_reordered, "tuple")
    if not refs.ok or not _rows_fit(be, refs, 1) or modules_hooked(refs.bypassed):
        return False
    pf, ps = past_key_value
    if not (torch.is_tensor(pf) and torch.is_tensor(ps) and pf.dim() == 4 and ps.dim() == 4 and pf.shape[0] == 2
            and ps.shape[0] == 2 and pf.shape[2] > 0 and pf.dtype == torch.bfloat16 and ps.dtype == torch.bfloat16
            and pf.device == hidden_states.device and ps.device == hidden_states.device):
        return False
    if ps.numel() and (ps.stride(3) != 1 or ps.data_ptr() % 16 or any(st % 8 for st in ps.stride()[:3])):
        return False
    attn = refs.attn
    if not hasattr(attn, "full_attention_heads") or "sink_size" not in attn.__dict__:
        return False
    _, num_kv, head_dim, _ = refs.dims
    if head_dim != 128 or pf.shape[3] != head_dim or pf.shape[1] + ps.shape[1] != num_kv:
        return False
    for t in position_embeddings:
        if not (torch.is_tensor(t) and t.dtype == torch.bfloat16 and t.is_cuda and t.numel() == head_dim
                and t.is_contiguous() and t.data_ptr() % 16 == 0):
            return False
    return True


def tuple_decode_attention_by_views(be, q, out, groups, nf, arena, N, past_stream, k, v, scale):
    """the fused tuple step's attention through the generic ``attention`` backend call (segments as tensor views): what
    ``tuple_decode_attention`` computes, for backends without it (the oracle) and for tests that record every call"""
    ns = k.shape[0] - nf
    full = (nf, 0, (arena[0, :, :N].transpose(0, 1), arena[1, :, :N].transpose(0, 1)),
            (k[:nf].unsqueeze(0), v[:nf].unsqueeze(0))) if nf > 0 else None
    stream = (ns, nf * groups,
              (past_stream[0].transpose(0, 1), past_stream[1].transpose(0, 1)) if past_stream.shape[2] > 0 else None,
              (k[nf:].unsqueeze(0), v[nf:].unsqueeze(0))) if ns > 0 else None
    be.attention(q.unsqueeze(0), out.unsqueeze(0), groups, full, stream, scale)


def duo_tuple_decode_layer_fused(layer, hidden_states, past_key_value, position_embeddings):
    """The tuple-cache decoder layer at q_len == 1 (reference tuple_kv_cache.py:431-490 around llama.py:146-306) in seven
    launches instead of ~45 torch kernels:

        q|k|v = [Wq; Wk; Wv] . rmsnorm(h)          input_layernorm in the projection's prologue (HF's two-rounding form)
        prep                                        HF rotary on q, k in place; retrieval rows appended to the arena;
                                                    the new streaming cache = truncate(old ++ new row), out of place
        attn  = decode attention                    retrieval heads over arena[:N] ++ new row, streaming heads over
                                                    old cache ++ new row (scan + merge launch)
        h1    = Wo . attn + h ;  g|u = [Wg; Wu] . rmsnorm(h1) ;  h2 = Wd . (silu(g) * u) + h1

    Returns ``(hidden_states, (full_KV, streaming_KV))`` with the tuple format's shapes: ``[2, nf, N + 1, D]`` — a view of
    the module-owned arena, as ``_tuple_full_kv_append`` hands out — and ``[2, ns, min(n + 1, sink + recent), D]`` (new)."""
    be = get_backend()
    r = _fused_refs(layer, duo_attention_forward_one_way_reordered, "tuple")
    attn = r.attn
    hidden = hidden_states.shape[2]
    num_heads, num_kv, head_dim, groups = r.dims
    x = hidden_states.reshape(1, hidden)
    qkv = be.token_linear(x, r.qkv, norm=(r.n_w, r.n_eps), **_norm_kw(r.n_hf))
    nq, nk = num_heads * head_dim, num_kv * head_dim
    q = qkv[0, :nq].view(num_heads, head_dim)
    k = qkv[0, nq:nq + nk].view(num_kv, head_dim)
    v = qkv[0, nq + nk:].view(num_kv, head_dim)
    ad = attn.__dict__
    if ad.get("full_attn_head_mask") is None:
        _tuple_head_split(attn, num_heads, num_kv, groups)
    nf, ns = ad["num_full_attn_head"], ad["num_streaming_attn_head"]
    past_full, past_stream = past_key_value
    arena, N = _tuple_arena_for(attn, past_full, 1, 1, nf, head_dim, hidden_states.device, hidden_states.dtype)
    buf = arena["buf"]
    cos, sin = position_embeddings
    new_stream = be.tuple_decode_prep(q, k, v, cos.view(-1), sin.view(-1), nf, buf, N, past_stream,
                                      ad["sink_size"], ad["recent_size"])
    arena["len"] = N + 1
    out = torch.empty(num_heads, head_dim, dtype=hidden_states.dtype, device=hidden_states.device)
    if hasattr(be, "tuple_decode_attention"):
        be.tuple_decode_attention(q, out, groups, nf, buf, N, past_stream, k, v, head_dim ** -0.5)
    else:
        tuple_decode_attention_by_views(be, q, out, groups, nf, buf, N, past_stream, k, v, head_dim ** -0.5)
    h1 = _out_linear(be, r.o, out.view(1, nq), None, x)
    gu = be.token_linear(h1, r.gu, norm=(r.p_w, r.p_eps), **_norm_kw(r.p_hf))
    h2 = _out_linear(be, r.down, gu[:, :r.inter], gu[:, r.inter:], h1)
    return h2.view(1, 1, hidden), (buf[:, :, :N + 1], new_stream)


# =============================================================================
# static dual-cache forward  (reference llama.py:309-434)
# =============================================================================
def duo_attention_forward_one_way_reordered_static(
    self,
    hidden_states: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    kv_cache: Optional[DuoAttentionStaticKVCache] = None,
    layer_idx: int = None,
    output_attentions: bool = False,
    use_cache: bool = False,
    pos0: Optional[int] = None,
    row_block: Optional[Tuple[int, int]] = None,
    **kwargs,
):
    """``row_block=(r0, chunk_len)``: ``hidden_states`` are rows [r0, r0 + q_len) of a prefill chunk of
    ``chunk_len`` rows (layer-pipeline wavefront, ``duo_static_attention_row_block``)."""
    bsz, q_len, _ = hidden_states.size()
    num_heads, num_kv, head_dim, groups = _dims(self)

    query_states = self.q_proj(hidden_states).view(bsz, q_len, num_heads, head_dim)
    key_states = self.k_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)
    value_states = self.v_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)

    rope_scale, rope_theta = rope_scale_and_theta(self.config)
    if row_block is not None:
        attn_output = duo_static_attention_row_block(query_states, key_states, value_states, kv_cache, layer_idx,
                                                     row_block[0], row_block[1], rope_scale, rope_theta)
    else:
        if pos0 is None and position_ids is not None:
            pos0 = first_positions(position_ids)
        attn_output = duo_static_attention_core(query_states, key_states, value_states, kv_cache, layer_idx,
                                                pos0, rope_scale, rope_theta)

    attn_output = attn_output.reshape(bsz, q_len, num_heads * head_dim)
    attn_output = self.o_proj(attn_output)
    return attn_output, None


# =============================================================================
# tuple-cache forward  (reference llama.py:146-306)
#   past_key_value = (full_KV [2B, nf, N, D], streaming_KV [2B, ns, <=W, D])
# =============================================================================
def duo_attention_forward_one_way_reordered(
    self,
    hidden_states: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_value: Optional[Tuple[torch.Tensor]] = None,
    output_attentions: bool = False,
    use_cache: bool = False,
    position_embeddings: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
    **kwargs,
):
    bsz, q_len, _ = hidden_states.size()
    num_heads, num_kv, head_dim, groups = _dims(self)

    query_states = self.q_proj(hidden_states).view(bsz, q_len, num_heads, head_dim)
    key_states = self.k_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)
    value_states = self.v_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)

    kv_seq_len = q_len
    if past_key_value is not None:
        kv_seq_len += past_key_value[0].shape[2]

    # HF rotary, as the reference's tuple path (:177-184)
    cos, sin = position_embeddings
    query_states, key_states = tuple_rotary(query_states, key_states, cos, sin)

    if not hasattr(self, "full_attn_head_mask") or self.full_attn_head_mask is None:
        _tuple_head_split(self, num_heads, num_kv, groups)
    nf, ns = self.num_full_attn_head, self.num_streaming_attn_head

    full_key_states = key_states[:, :, :nf, :]
    full_value_states = value_states[:, :, :nf, :]
    streaming_key_states = key_states[:, :, nf:, :]
    streaming_value_states = value_states[:, :, nf:, :]

    be = get_backend()
    attn_output = torch.empty_like(query_states)
    scale = head_dim ** -0.5
    batched = bsz > 1 and hasattr(be, "attention_batched")      # one launch for all batch rows
    if past_key_value is None or q_len == kv_seq_len:
        if batched:
            be.attention_batched(query_states, attn_output, groups, (num_kv, 0, None, (key_states, value_states)), None, scale)
        for b in range(0 if batched else bsz):
            be.attention(query_states[b], attn_output[b], groups,
                         (num_kv, 0, None, (key_states[b], value_states[b])), None, scale)
        if past_key_value is not None:  # zero-length past: keep the concatenations well-formed
            past_full_KV = past_key_value[0].transpose(1, 2)
            past_streaming_KV = past_key_value[1].transpose(1, 2)
        else:
            past_full_KV = past_streaming_KV = None
    else:
        past_full_KV = past_key_value[0].transpose(1, 2)        # [2B, N, nf, D]
        past_streaming_KV = past_key_value[1].transpose(1, 2)   # [2B, n, ns, D]
        if batched:     # K rows = the first B entries of the stacked cache, V rows = the last B
            full = (nf, 0, (past_full_KV[:bsz], past_full_KV[bsz:]), (full_key_states, full_value_states)) if nf > 0 else None
            stream = (ns, nf * groups, (past_streaming_KV[:bsz], past_streaming_KV[bsz:]),
                      (streaming_key_states, streaming_value_states)) if ns > 0 else None
            be.attention_batched(query_states, attn_output, groups, full, stream, scale)
        for b in range(0 if batched else bsz):
            full = (nf, 0, (past_full_KV[b], past_full_KV[bsz + b]),
                    (full_key_states[b], full_value_states[b])) if nf > 0 else None
            stream = (ns, nf * groups, (past_streaming_KV[b], past_streaming_KV[bsz + b]),
                      (streaming_key_states[b], streaming_value_states[b])) if ns > 0 else None
            be.attention(query_states[b], attn_output[b], groups, full, stream, scale)

    # the tuple format's contract: the returned cache holds past ++ new (:202-223).  The streaming part is
    # at most sink+recent+q_len rows: concatenated as in the reference.  The retrieval part is the whole
    # context — re-concatenating it costs O(N) reads and writes per generated token (twice: K/V cat, then the
    # K-on-V stack), several times the attention's own traffic at 128K.  Here it lives in a growing arena
    # owned by the module and the tuple element handed back is a VIEW of it: linear generation appends q_len
    # rows in place.  A past that is not the arena's current view (first call, a re-used older tuple, a
    # foreign tensor) is copied into a fresh arena, so branching callers stay correct.
    if past_streaming_KV is not None:
        streaming_key_states = torch.cat([past_streaming_KV[:bsz], streaming_key_states], dim=1)
        streaming_value_states = torch.cat([past_streaming_KV[bsz:], streaming_value_states], dim=1)
    full_kv_out = None
    if not use_cache and past_key_value is None:
        release_tuple_arena(self)     # a cache-less call: nothing will come back for the arena
    if use_cache:
        full_kv_out = _tuple_full_kv_append(self, past_key_value[0] if past_key_value is not None else None,
                                            full_key_states, full_value_states)

    attn_output = attn_output.reshape(bsz, q_len, num_heads * head_dim)
    attn_output = self.o_proj(attn_output)

    # sink + recent truncation (:273-290)
    if streaming_key_states.shape[1] > self.recent_size + self.sink_size:
        W = self.sink_size + self.recent_size
        streaming_key_states = torch.cat(
            [streaming_key_states[:, : self.sink_size], streaming_key_states[:, -self.recent_size:]], dim=1
        )[:, :W]
        streaming_value_states = torch.cat(
            [streaming_value_states[:, : self.sink_size], streaming_value_states[:, -self.recent_size:]], dim=1
        )[:, :W]

    past_key_value = (
        (
            full_kv_out,
            torch.cat([streaming_key_states, streaming_value_states], dim=0).transpose(1, 2),
        )
        if use_cache
        else None
    )
    return attn_output, None, past_key_value


def _tuple_head_split(module, num_heads, num_kv, groups):
    """head counts of the two classes from the registered pattern, computed once (reference llama.py:186-195; one
    device read-back)"""
    module.full_attn_head_mask = module.full_attention_heads > 0.5
    module.num_full_attn_head = int(module.full_attn_head_mask.sum().item())
    module.num_streaming_attn_head = num_kv - module.num_full_attn_head
    module.num_full_query_head = module.num_full_attn_head * groups
    module.num_streaming_query_head = num_heads - module.num_full_query_head


def release_tuple_arena(module_or_model):
    """Free the retrieval-head arena(s) of the tuple path (up to 1.5x a layer's full KV each).  The arena lives on
    the attention module between calls so that linear generation appends in place; call this (on a module or on
    the whole model) when a sequence is finished and its ``past_key_values`` are dropped.  Tuples handed back
    earlier stay valid: they are views that keep their storage alive."""
    mods = module_or_model.modules() if hasattr(module_or_model, "modules") else [module_or_model]
    for m in mods:
        if getattr(m, "_duo_full_kv_arena", None) is not None:
            m._duo_full_kv_arena = None


def _tuple_full_kv_append(module, past_full, new_k, new_v):
    """Retrieval-head cache of the tuple format, ``[2B, nf, N, D]`` (K stacked on V, head-major,
    reference llama.py:168-171,292-301), grown in place.  ``past_full``: the tuple element of the previous
    call or None; ``new_k/new_v``: ``[B, q, nf, D]``.  Returns the ``[2B, nf, N+q, D]`` view.

    Aliasing contract (differs from the reference's fresh ``torch.cat``): the returned tensor is a VIEW of the
    module-owned arena.  Its own rows are never rewritten, but the storage behind it grows in place, and a
    caller that alternates two sequences on one model makes every call copy (the arena follows the last
    sequence).  ``release_tuple_arena`` frees it."""
    bsz, q, nf, D = new_k.shape
    arena, N = _tuple_arena_for(module, past_full, bsz, q, nf, D, new_k.device, new_k.dtype)
    buf = arena["buf"]
    buf[:bsz, :, N:N + q].copy_(new_k.transpose(1, 2))
    buf[bsz:, :, N:N + q].copy_(new_v.transpose(1, 2))
    arena["len"] = N + q
    return buf[:, :, :N + q]


def _tuple_arena_for(module, past_full, bsz, q, nf, D, device, dtype):
    """the module's arena with room for ``q`` more rows behind ``past_full`` (None or ``[2B, nf, N, D]``): the current one
    when ``past_full`` IS its current view, else a fresh one holding a copy of ``past_full``.  Returns (arena, N); the
    caller writes rows [N, N + q) and sets ``arena["len"]``."""
    N = 0 if past_full is None else past_full.shape[2]
    arena = getattr(module, "_duo_full_kv_arena", None)
    fits = (
        arena is not None and past_full is not None and arena["len"] == N
        and past_full.data_ptr() == arena["buf"].data_ptr() and past_full.shape[:2] == arena["buf"].shape[:2]
        and past_full.stride() == arena["buf"].stride() and N + q <= arena["buf"].shape[2]
        and past_full.dtype == dtype
    )
    if not fits:
        cap = N + q + max(1024, (N + q) // 2)       # 1.5x growth: amortised O(1) copies per appended row
        buf = torch.empty(2 * bsz, nf, cap, D, device=device, dtype=dtype)
        if N > 0:
            buf[:, :, :N].copy_(past_full)
        arena = {"buf": buf, "len": N}
        module._duo_full_kv_arena = arena
    return arena, N


# =============================================================================
# enablers
# =============================================================================
def _reorder_layer(module, layer_full_attention_heads):
    """q/k/v rows and o_proj columns permuted so retrieval heads come first (reference llama.py:523-546)."""
    _, _, head_dim, groups = _dims(module)
    # which original kv head sits at which position now (a later head-parallel split of the ALREADY reordered model —
    # reference harness order: enabler, then to_device(enable_tp=True), eval/needle/needle_in_haystack.py:195-214 —
    # has to translate a whole-model pattern given in the original head order): current position -> original head id
    mask = (torch.as_tensor(layer_full_attention_heads).float() > 0.5).tolist()
    perm = [h for h, m in enumerate(mask) if m] + [h for h, m in enumerate(mask) if not m]
    prev = module.__dict__.get("_duo_head_order")
    module._duo_head_order = [prev[p] for p in perm] if prev else perm
    module.q_proj = reorder_linear_weights(module.q_proj, layer_full_attention_heads, groups * head_dim, "out")
    module.k_proj = reorder_linear_weights(module.k_proj, layer_full_attention_heads, head_dim, "out")
    module.v_proj = reorder_linear_weights(module.v_proj, layer_full_attention_heads, head_dim, "out")
    module.o_proj = reorder_linear_weights(module.o_proj, layer_full_attention_heads, groups * head_dim, "in")


def _layer_rows(model, full_attention_heads):
    """the head-pattern rows of the layers this process holds (all of them, or a pipeline stage's block)"""
    pp = getattr(model, "_duo_pp", None)
    if pp is not None:
        full_attention_heads = pp.local_rows(full_attention_heads)
    if getattr(model, "_duo_tp", None) is not None:      # head-parallel shard: this rank's heads of every row
        from ..tp import tp_local_rows

        full_attention_heads = tp_local_rows(model, full_attention_heads)
    return full_attention_heads


def enable_duo_attention_eval(model, full_attention_heads, sink_size, recent_size):
    """Tuple-cache eval patch (reference llama.py:504-554)."""
    enable_tuple_kv_cache_for_model(model)
    device = next(model.parameters()).device
    dtype = next(model.parameters()).dtype
    if getattr(model, "_duo_tp", None) is not None:
        from ..tp import tp_local_rows

        full_attention_heads = tp_local_rows(model, full_attention_heads)
    for idx, layer in enumerate(model.model.layers):
        module = layer.self_attn
        layer_heads = torch.as_tensor(full_attention_heads[idx]).to(device=device, dtype=dtype)
        module.forward = types.MethodType(duo_attention_forward_one_way_reordered, module)
        _reorder_layer(module, layer_heads)
        layer_heads = reorder_full_attn_heads(layer_heads)
        module.sink_size = sink_size
        module.recent_size = recent_size
        module.full_attn_head_mask = None
        module.register_buffer("full_attention_heads", layer_heads)


def enable_duo_attention_static_kv_cache_eval(model, full_attention_heads):
    """Static-cache eval patch (reference llama.py:557-598)."""
    enable_duo_attention_static_kv_cache(model)
    enable_flashinfer_rmsnorm(model)
    device = next(model.parameters()).device
    dtype = next(model.parameters()).dtype
    full_attention_heads = _layer_rows(model, full_attention_heads)
    for idx, layer in enumerate(model.model.layers):
        module = layer.self_attn
        layer_heads = torch.as_tensor(full_attention_heads[idx]).to(device=device, dtype=dtype)
        module.forward = types.MethodType(duo_attention_forward_one_way_reordered_static, module)
        _reorder_layer(module, layer_heads)


def enable_duo_attention_training(model, sink_size, recent_size, max_length, initial_value=1.0,
                                  enable_ulysses_attention=False, streaming_attn_implementation="blocksparse"):
    raise NotImplementedError(
        "Retrieval-head identification (training, reference llama.py:437-501) is outside the MI355X "
        "hot-path scope; use the shipped attn_patterns."
    )


def _attn_modules(model):
    inner = model.model if hasattr(model, "model") and hasattr(model.model, "layers") else model
    if not hasattr(inner, "layers"):
        raise ValueError("Model type not supported")
    for layer in inner.layers:
        yield layer.self_attn


def get_full_attention_heads(model):
    """reference llama.py:601-640.  Single-process model: the per-layer buffers.  A model sharded with
    ``duo_attn.tp.shard_model_for_tp``: the reference's TensorParallel branch (:603-622) — every rank gets the whole
    model's per-layer ``[Hkv]`` rows (original kv-head order), gathered over the TP group."""
    local = [m.full_attention_heads for m in _attn_modules(model) if hasattr(m, "full_attention_heads")]
    if getattr(model, "_duo_tp", None) is not None and local:
        from ..tp import gather_full_attention_heads

        return gather_full_attention_heads(model, local)
    return local


def set_full_attention_heads(model, full_attention_heads):
    """reference llama.py:643-672; on a TP-sharded model the rows are whole-model ``[Hkv]`` rows and each rank
    keeps its own heads."""
    tp = getattr(model, "_duo_tp", None)
    for layer_idx, m in enumerate(_attn_modules(model)):
        if not hasattr(m, "full_attention_heads"):
            continue
        row = full_attention_heads[layer_idx]
        if tp is not None and row.numel() == tp["num_kv_heads"]:
            from ..tp import scatter_full_attention_heads

            row = scatter_full_attention_heads(model, layer_idx, row)
        m.full_attention_heads.data = row.to(m.full_attention_heads.device, m.full_attention_heads.dtype)
        m.full_attn_head_mask = None
    return model


def map_full_attention_heads(model, func):
    """reference llama.py:675-693 (applied to this process's buffers: all of them, or a TP rank's slice)"""
    for m in _attn_modules(model):
        if hasattr(m, "full_attention_heads"):
            func(m.full_attention_heads)
