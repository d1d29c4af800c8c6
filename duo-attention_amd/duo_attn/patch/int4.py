"""HF-model glue for the INT4 dual KV cache (BASELINE config 5).

Mirror of the attention forward of the reference's QServe demo model (``demo/w8a8kv4_llama.py:174-287``) on a
HuggingFace Llama/Mistral in **fp16**: q/k/v projections -> RoPE in place -> ``kv_cache.put`` (quantise the new rows
into the retrieval and streaming pools) -> first chunk: every head causal over the chunk's own K/V; later chunks:
retrieval heads over the dequantised full pool, streaming heads over the dequantised streaming pool; a decode
step: attention straight on the packed pools -> ``kv_cache.compress``.  The demo's W8A8 linears and fused
activation quantisation are third-party QServe kernels and stay out of scope: the projections and the
MLP here are the model's own fp16 ``nn.Linear`` (hipBLASLt).

    model = LlamaForCausalLM(...).half().cuda()
    enable_llama_duo_attention_int4_kv_eval(model, full_attention_heads)
    kv = DuoAttentionStaticINT4KVCache(model, full_attention_heads, 1, max_size, sink, recent, chunk)
    out = model(input_ids=chunk_ids, past_key_values=kv, use_cache=True)     # same calls as the bf16 static path
"""
from __future__ import annotations

import types
from typing import Optional

import torch

from ..int4_kv import DuoAttentionStaticINT4KVCache
from ._duo import _dims, _layer_rows, _reorder_layer, first_positions, rope_scale_and_theta
from .flashinfer_utils import apply_rope_inplace
from .static_kv_cache import enable_duo_attention_static_kv_cache


def duo_attention_forward_int4_kv(
    self,
    hidden_states: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    kv_cache: Optional[DuoAttentionStaticINT4KVCache] = None,
    layer_idx: int = None,
    output_attentions: bool = False,
    use_cache: bool = False,
    pos0=None,
    **kwargs,
):
    """reference demo/w8a8kv4_llama.py:174-287 (attention part), fp16."""
    bsz, q_len, _ = hidden_states.size()
    num_heads, num_kv, head_dim, groups = _dims(self)
    if hidden_states.dtype != torch.float16:
        raise ValueError("the INT4-KV path is fp16 like the reference's QServe model; call model.half() first")

    query_states = self.q_proj(hidden_states).view(bsz, q_len, num_heads, head_dim)
    key_states = self.k_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)
    value_states = self.v_proj(hidden_states).view(bsz, q_len, num_kv, head_dim)

    past = kv_cache.kv_seq_len                       # last layer's counter, as the reference reads it (:201-203)
    if pos0 is None:
        pos0 = first_positions(position_ids) if position_ids is not None else past
    rope_scale, rope_theta = rope_scale_and_theta(self.config)
    apply_rope_inplace(query_states, key_states, pos0, rope_scale, rope_theta)      # :206-215

    kv_cache.put(layer_idx, key_states, value_states, dequantize=False)             # :225-230
    if q_len == 1 and past > 0:
        attn_output = kv_cache.decode_attention(layer_idx, query_states)            # :240-274 on the packed pools
    else:
        attn_output = kv_cache.prefill_attention(layer_idx, query_states, key_states, value_states)   # :232-274
    kv_cache.compress(layer_idx)                                                    # :278

    attn_output = attn_output.reshape(bsz, q_len, num_heads * head_dim)
    return self.o_proj(attn_output), None


def enable_duo_attention_int4_kv_eval(model, full_attention_heads):
    """Patch a HF Llama/Mistral (fp16) for the INT4 dual cache: static-cache model/layer/LM forwards (they thread
    ``kv_cache`` + ``layer_idx`` to the attention, reference static_kv_cache.py:318-567), retrieval-first weight
    reorder (llama.py:523-546), and the INT4 attention forward above.  The model's own RMSNorm is kept (the HIP
    RMSNorm kernel is bf16)."""
    enable_duo_attention_static_kv_cache(model)
    device = next(model.parameters()).device
    dtype = next(model.parameters()).dtype
    if dtype != torch.float16:
        raise ValueError(f"the INT4-KV path is fp16 (reference demo/w8a8kv4_llama.py); the model is {dtype}")
    rows = _layer_rows(model, full_attention_heads)
    for idx, layer in enumerate(model.model.layers):
        module = layer.self_attn
        layer_heads = torch.as_tensor(rows[idx]).to(device=device, dtype=dtype)
        module.forward = types.MethodType(duo_attention_forward_int4_kv, module)
        _reorder_layer(module, layer_heads)


enable_llama_duo_attention_int4_kv_eval = enable_duo_attention_int4_kv_eval
enable_mistral_duo_attention_int4_kv_eval = enable_duo_attention_int4_kv_eval
