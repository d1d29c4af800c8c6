"""Tuple-style KV cache forwards for transformers 5.x models.

Mirror of the reference's ``duo_attn/patch/tuple_kv_cache.py``: it re-implements
the HF-4.34 ``ForCausalLM / Model / DecoderLayer`` forwards with per-layer tuple
caches for llama (:241-510) and mistral (:514-783, identical), plus the
full-attention baseline ``old_flash_attention_2_forward`` (:38-120).  Those old
forwards cannot run on transformers 5.x (``DynamicCache``, ``position_embeddings``,
no per-layer ``rotary_emb``), so the same data flow is re-targeted here:

    out = model(input_ids=[B,S], past_key_values=None | tuple, use_cache=True)
    out.logits            [B, 1, V] fp32 — last position only in eval (:283-288)
    out.past_key_values   tuple(layer -> per-layer cache tuple)

The unpad/varlen helpers of the reference (:24-35,123-237) are dead code there
(``torch.functional`` has no ``pad``) and are not reproduced; B=1 or
equal-length rows only, like every reference harness.
"""
from __future__ import annotations

import types
from typing import Optional, Tuple

import torch
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from ..backend import get_backend


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def hf_apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=1):
    """HF's rotary application (bf16 arithmetic on bf16 cos/sin), used by the tuple path exactly
    as the reference does (llama.py:177-184)."""
    cos = cos.unsqueeze(unsqueeze_dim)
    sin = sin.unsqueeze(unsqueeze_dim)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def tuple_rotary(q, k, cos, sin):
    """the tuple forwards' rotary call (reference llama.py:177-184: ``apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=2)``)
    on q [B, S, Hq, D] / k [B, S, Hkv, D] fresh from the projections: ONE in-place pass per batch row on the GPU
    (``duo_rope_hf_inplace_bf16``, bit-equal to the torch sequence), the torch sequence itself anywhere else"""
    be = get_backend()

    def _aligned(t):        # what duo_rope_hf_inplace_bf16 accepts: 16-byte aligned rows (base and every stride in elements % 8)
        return t.data_ptr() % 16 == 0 and all(st % 8 == 0 for st in t.stride()[:-1])

    if (hasattr(be, "rope_hf_inplace") and q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16
            and cos.dtype == torch.bfloat16 and sin.dtype == torch.bfloat16 and cos.dim() == 3 and sin.shape == cos.shape
            and q.dim() == 4 and k.dim() == 4 and q.shape[-1] == 128 and q.stride(-1) == 1 and k.stride(-1) == 1
            and cos.shape[0] in (1, q.shape[0]) and cos.is_contiguous() and sin.is_contiguous()
            and _aligned(q) and _aligned(k) and _aligned(cos) and _aligned(sin)     # (a sliced cached table: torch sequence)
            and not (q.requires_grad or k.requires_grad)):
        for b in range(q.shape[0]):
            cb = cos[b if cos.shape[0] > 1 else 0]
            sb = sin[b if sin.shape[0] > 1 else 0]
            be.rope_hf_inplace(q[b], k[b], cb, sb)
        return q, k
    return hf_apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=2)


def _hf_norm(norm, x):
    """``norm(x)`` for a HuggingFace *RMSNorm module that still runs its own forward: one pass on the GPU
    (``duo_rmsnorm_hf_bf16``: the module's two-rounding arithmetic), the module itself anywhere else"""
    be = get_backend()
    if (hasattr(be, "rmsnorm_hf") and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0
            and not x.requires_grad):
        from ._duo import _norm_form, modules_hooked

        if _norm_form(norm) == "hf" and not modules_hooked((norm,)):
            return be.rmsnorm_hf(x, norm.weight, norm.variance_epsilon)
    return norm(x)


def _past_length(past_key_values) -> int:
    if past_key_values is None:
        return 0
    return past_key_values[0][0].shape[2]


# ----------------------------------------------------------------------------
# full-attention baseline on a (K, V) tuple cache  (reference :38-120)
#   past_key_value = (K [B, Hkv, N, D], V [B, Hkv, N, D])
# ----------------------------------------------------------------------------
def tuple_full_attention_forward(
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
    cfg = self.config
    num_heads, num_kv = cfg.num_attention_heads, cfg.num_key_value_heads
    head_dim = getattr(self, "head_dim", cfg.hidden_size // num_heads)
    groups = num_heads // num_kv

    q = self.q_proj(hidden_states).view(bsz, q_len, num_heads, head_dim)
    k = self.k_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)
    v = self.v_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)
    cos, sin = position_embeddings
    q, k = tuple_rotary(q, k, cos, sin)

    be = get_backend()
    out = torch.empty_like(q)
    scale = head_dim ** -0.5
    for b in range(bsz):
        segA = None
        if past_key_value is not None and past_key_value[0].shape[2] > 0:
            segA = (past_key_value[0][b].transpose(0, 1), past_key_value[1][b].transpose(0, 1))
        be.attention(q[b], out[b], groups, (num_kv, 0, segA, (k[b], v[b])), None, scale)

    k_t, v_t = k.transpose(1, 2), v.transpose(1, 2)
    if past_key_value is not None:
        k_t = torch.cat([past_key_value[0], k_t], dim=2)
        v_t = torch.cat([past_key_value[1], v_t], dim=2)
    new_cache = (k_t, v_t) if use_cache else None

    out = out.reshape(bsz, q_len, num_heads * head_dim)
    return self.o_proj(out), None, new_cache


def tuple_for_causal_lm_forward(
    self,
    input_ids: torch.LongTensor = None,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_values=None,
    inputs_embeds: Optional[torch.FloatTensor] = None,
    labels: Optional[torch.LongTensor] = None,
    use_cache: Optional[bool] = None,
    **kwargs,
):
    outputs = self.model(
        input_ids=input_ids,
        attention_mask=attention_mask,
        position_ids=position_ids,
        past_key_values=past_key_values,
        inputs_embeds=inputs_embeds,
        use_cache=use_cache,
    )
    hidden_states = outputs.last_hidden_state
    if self.training:
        logits = self.lm_head(hidden_states)
    else:
        logits = self.lm_head(hidden_states[:, -1:, :])
    logits = logits.float()
    loss = None
    if labels is not None:
        shift_logits = logits[..., :-1, :].contiguous().view(-1, self.config.vocab_size)
        shift_labels = labels[..., 1:].contiguous().view(-1).to(shift_logits.device)
        loss = torch.nn.functional.cross_entropy(shift_logits, shift_labels)
    return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=outputs.past_key_values)


def tuple_model_forward(
    self,
    input_ids: torch.LongTensor = None,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_values=None,
    inputs_embeds: Optional[torch.FloatTensor] = None,
    use_cache: Optional[bool] = None,
    **kwargs,
):
    if input_ids is not None and inputs_embeds is not None:
        raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
    if input_ids is not None:
        _, seq_length = input_ids.shape
    elif inputs_embeds is not None:
        _, seq_length, _ = inputs_embeds.shape
    else:
        raise ValueError("You have to specify either input_ids or inputs_embeds")
    use_cache = use_cache if use_cache is not None else self.config.use_cache

    past_len = _past_length(past_key_values)   # reference :358-362
    if position_ids is None:
        device = input_ids.device if input_ids is not None else inputs_embeds.device
        position_ids = torch.arange(past_len, seq_length + past_len, dtype=torch.long, device=device)
        position_ids = position_ids.unsqueeze(0).view(-1, seq_length)
    else:
        position_ids = position_ids.view(-1, seq_length).long()

    if inputs_embeds is None:
        inputs_embeds = self.embed_tokens(input_ids)
    hidden_states = inputs_embeds
    position_embeddings = self.rotary_emb(hidden_states, position_ids)

    next_cache = () if use_cache else None
    for idx, decoder_layer in enumerate(self.layers):
        pkv = past_key_values[idx] if past_key_values is not None else None
        hidden_states, layer_cache = decoder_layer(
            hidden_states,
            position_ids=position_ids,
            past_key_value=pkv,
            use_cache=use_cache,
            position_embeddings=position_embeddings,
        )
        if use_cache:
            next_cache += (layer_cache,)
    hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=next_cache)


def tuple_decoder_layer_forward(
    self,
    hidden_states: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_value=None,
    output_attentions: Optional[bool] = False,
    use_cache: Optional[bool] = False,
    position_embeddings=None,
    **kwargs,
):
    if hidden_states.shape[1] == 1 and past_key_value is not None:
        # decode step: the token-row linears fused around the attention op, the cache updates in one launch
        # (duo_attn/patch/_duo.py: duo_tuple_decode_layer_fused; anything it does not cover takes the module sequence below)
        from ._duo import duo_tuple_decode_layer_fused, tuple_fused_decode_ok

        if tuple_fused_decode_ok(self, hidden_states, past_key_value, position_embeddings, use_cache):
            return duo_tuple_decode_layer_fused(self, hidden_states, past_key_value, position_embeddings)
    # (chunks: HF's norm arithmetic and the SwiGLU activation product as single passes on the GPU — `_hf_norm`,
    #  static_kv_cache._mlp_forward; the modules themselves wherever those do not apply)
    from .static_kv_cache import _mlp_forward

    residual = hidden_states
    hidden_states = _hf_norm(self.input_layernorm, hidden_states)
    hidden_states, _, present = self.self_attn(
        hidden_states=hidden_states,
        position_ids=position_ids,
        past_key_value=past_key_value,
        use_cache=use_cache,
        position_embeddings=position_embeddings,
    )
    hidden_states = residual + hidden_states
    residual = hidden_states
    hidden_states = _hf_norm(self.post_attention_layernorm, hidden_states)
    hidden_states = _mlp_forward(self.mlp, hidden_states)
    hidden_states = residual + hidden_states
    return hidden_states, present


def enable_tuple_kv_cache_for_model(model):
    """Rebind LM / model / layer / attention forwards to the tuple-cache versions (reference :493-510)."""
    model.model.forward = types.MethodType(tuple_model_forward, model.model)
    for layer in model.model.layers:
        layer.forward = types.MethodType(tuple_decoder_layer_forward, layer)
        layer.self_attn.forward = types.MethodType(tuple_full_attention_forward, layer.self_attn)
    model.forward = types.MethodType(tuple_for_causal_lm_forward, model)


enable_tuple_kv_cache_for_llama = enable_tuple_kv_cache_for_model
enable_tuple_kv_cache_for_mistral = enable_tuple_kv_cache_for_model


def enable_tuple_kv_cache(model):
    """reference :786-792"""
    mt = model.config.model_type
    if "llama" in mt or "mistral" in mt or "mixtral" in mt:
        enable_tuple_kv_cache_for_model(model)
    else:
        raise ValueError(f"Model type {mt} not supported")
