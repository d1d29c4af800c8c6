"""Static dual KV cache + the static-cache model forwards.

Host-side mirror of the reference's ``duo_attn/patch/static_kv_cache.py``
(``DuoAttentionStaticKVCache`` :18-315, model/layer/LM forwards :318-567 for
llama and :570-820 for mistral — the two families are textually identical
there, so one implementation serves both).

MI355X-first differences that do not change the interface:
  * pools are allocated HEAD-major ``[B, h, T, D]`` in HBM and exposed as the
    reference's token-major *shape* ``[B, T, h, D]`` through a strided view, so
    each kv head is one contiguous 256-B-row stream for the decode scan and one
    contiguous 16-KiB block per 64-key MFMA tile; every slicing / ``copy_`` the
    reference's callers do on these tensors still works;
  * ``put_full_kv`` / ``compress_and_replace_streaming_kv`` move rows with HIP
    kernels (``duo_kv_append_bf16`` / ``duo_stream_compress_bf16``) instead of
    chains of ``copy_``; the streaming update takes the *new* rows and the pool
    (two segments) — the reference's ``torch.cat`` is never materialised.
"""
from __future__ import annotations

import types
from typing import Optional

import torch
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from ..backend import get_backend


class DuoAttentionStaticKVCache:
    """Same constructor, attributes and methods as reference static_kv_cache.py:18-315."""

    def __init__(self, model, full_attention_heads, batch_size, max_size, sink_size, recent_size):
        self.batch_size = batch_size
        self.max_size = max_size
        self.sink_size = sink_size
        self.recent_size = recent_size

        self.device = next(model.parameters()).device
        self.dtype = next(model.parameters()).dtype
        # a model sharded over pipeline stages (duo_attn.pipeline.shard_model_for_pp): this rank owns the pools of
        # the layers it kept; `full_attention_heads` may be the whole model's pattern or already the stage's rows
        pp = getattr(model, "_duo_pp", None)
        if pp is not None:
            full_attention_heads = pp.local_rows(full_attention_heads)
        if getattr(model, "_duo_tp", None) is not None:     # head-parallel shard: a whole-model pattern is sliced to this rank's heads
            from ..tp import tp_local_rows

            full_attention_heads = tp_local_rows(model, full_attention_heads)
        self.num_layers = len(full_attention_heads) if pp is not None else model.config.num_hidden_layers
        self.num_heads = model.config.num_attention_heads
        self.num_kv_heads = model.config.num_key_value_heads
        self.num_kv_groups = self.num_heads // self.num_kv_heads
        # (an explicit config.head_dim wins over hidden_size // num_heads — they differ in some checkpoints
        # and in a tensor-parallel shard, where the head count is per rank)
        self.head_dim = getattr(model.config, "head_dim", None) or model.config.hidden_size // self.num_heads

        self.num_full_kv_head_list = [0] * self.num_layers
        self.num_streaming_kv_head_list = [0] * self.num_layers
        self.kv_seq_len_list = [0] * self.num_layers
        self.streaming_kv_seq_len_list = [0] * self.num_layers

        self.streaming_key_states_list = []
        self.streaming_value_states_list = []
        self.full_key_states_list = []
        self.full_value_states_list = []

        W = self.sink_size + self.recent_size
        for idx, layer_heads in enumerate(full_attention_heads):
            mask = torch.as_tensor(layer_heads) > 0.5
            nf = int(mask.sum().item())
            ns = self.num_kv_heads - nf
            self.num_full_kv_head_list[idx] = nf
            self.num_streaming_kv_head_list[idx] = ns
            self.streaming_key_states_list.append(self._alloc(W, ns))
            self.streaming_value_states_list.append(self._alloc(W, ns))
            self.full_key_states_list.append(self._alloc(self.max_size, nf))
            self.full_value_states_list.append(self._alloc(self.max_size, nf))

    def _alloc(self, rows, heads):
        # physical [B, h, T, D]; logical (reference) shape [B, T, h, D]
        buf = torch.zeros(self.batch_size, heads, rows, self.head_dim, device=self.device, dtype=self.dtype)
        return buf.permute(0, 2, 1, 3)

    # ------------------------------------------------------------------ counters
    @property
    def streaming_kv_seq_len(self):
        return self.streaming_kv_seq_len_list[-1]

    @property
    def kv_seq_len(self):
        return self.kv_seq_len_list[-1]

    # ------------------------------------------------------------------ full pool
    def put_full_kv(self, layer_idx, full_key_states, full_value_states):
        incoming = full_key_states.shape[1]
        cur = self.kv_seq_len_list[layer_idx]
        if incoming + cur > self.max_size:
            raise ValueError(
                f"Trying to put {incoming} KVs into a cache with max size {self.max_size}, current size: {cur}."
            )
        be = get_backend()
        kp, vp = self.full_key_states_list[layer_idx], self.full_value_states_list[layer_idx]
        nb = full_key_states.shape[0]
        if nb > 1 and hasattr(be, "kv_append_batched"):      # all batch rows in one launch
            be.kv_append_batched(full_key_states, full_value_states, kp, vp, cur)
            nb = 0
        for b in range(nb):
            be.kv_append(full_key_states[b], full_value_states[b], kp[b], vp[b], cur)
        self.kv_seq_len_list[layer_idx] += incoming
        return self.get_full_kv(layer_idx)

    def get_full_kv(self, layer_idx):
        n = self.kv_seq_len_list[layer_idx]
        return self.full_key_states_list[layer_idx][:, :n], self.full_value_states_list[layer_idx][:, :n]

    # ------------------------------------------------------------------ streaming pool
    def get_streaming_kv(self, layer_idx):
        n = self.streaming_kv_seq_len_list[layer_idx]
        return self.streaming_key_states_list[layer_idx][:, :n], self.streaming_value_states_list[layer_idx][:, :n]

    def compress_and_replace_streaming_kv(self, layer_idx, streaming_key_states, streaming_value_states):
        """Reference signature (static_kv_cache.py:127-167): the argument is the FULL logical
        streaming sequence (``torch.cat([cached, new])``).  Kept for drop-in callers; the patched
        forward uses :meth:`update_streaming_kv`, which never builds that concatenation."""
        incoming = streaming_key_states.shape[1]
        W = self.sink_size + self.recent_size
        be = get_backend()
        kp, vp = self.streaming_key_states_list[layer_idx], self.streaming_value_states_list[layer_idx]
        if kp.shape[2] > 0:
            # the input may alias the pool (the reference passes views of a fresh cat, never the
            # pool itself); treat it as an independent tensor: pool := compress(input)
            nb = streaming_key_states.shape[0]
            if nb > 1 and hasattr(be, "stream_compress_batched"):      # all batch rows in one launch
                be.stream_compress_batched(kp, vp, streaming_key_states, streaming_value_states, 0, self.sink_size,
                                           self.recent_size)
                nb = 0
            for b in range(nb):
                be.stream_compress(kp[b], vp[b], streaming_key_states[b], streaming_value_states[b], 0,
                                   self.sink_size, self.recent_size)
        self.streaming_kv_seq_len_list[layer_idx] = min(incoming, W)

    def update_streaming_kv(self, layer_idx, new_key_states, new_value_states):
        """pool := compress(pool[:len] ++ new) in one in-place kernel; same result as the reference's
        cat + compress_and_replace_streaming_kv pair (llama.py:385-390,423-425)."""
        cur = self.streaming_kv_seq_len_list[layer_idx]
        W = self.sink_size + self.recent_size
        be = get_backend()
        kp, vp = self.streaming_key_states_list[layer_idx], self.streaming_value_states_list[layer_idx]
        if kp.shape[2] > 0:
            nb = new_key_states.shape[0]
            if nb > 1 and hasattr(be, "stream_compress_batched"):      # all batch rows in one launch
                be.stream_compress_batched(kp, vp, new_key_states, new_value_states, cur, self.sink_size, self.recent_size)
                nb = 0
            for b in range(nb):
                be.stream_compress(kp[b], vp[b], new_key_states[b], new_value_states[b], cur,
                                   self.sink_size, self.recent_size)
        self.streaming_kv_seq_len_list[layer_idx] = min(cur + new_key_states.shape[1], W)

    # ------------------------------------------------------------------ views
    def get(self, layer_idx):
        return (*self.get_full_kv(layer_idx), *self.get_streaming_kv(layer_idx))

    def get_unsliced(self, layer_idx):
        return (
            self.kv_seq_len_list[layer_idx],
            self.full_key_states_list[layer_idx],
            self.full_value_states_list[layer_idx],
            self.streaming_kv_seq_len_list[layer_idx],
            self.streaming_key_states_list[layer_idx],
            self.streaming_value_states_list[layer_idx],
        )

    def split_kv(self, layer_idx, key_states, value_states):
        nf = self.num_full_kv_head_list[layer_idx]
        return (
            key_states[:, :, :nf, :],
            value_states[:, :, :nf, :],
            key_states[:, :, nf:, :],
            value_states[:, :, nf:, :],
        )

    def update_seq_len(self, layer_idx, incoming_kv_seq_len):
        self.kv_seq_len_list[layer_idx] += incoming_kv_seq_len
        self.streaming_kv_seq_len_list[layer_idx] += incoming_kv_seq_len

    def clear(self):
        for i in range(self.num_layers):
            self.kv_seq_len_list[i] = 0
            self.streaming_kv_seq_len_list[i] = 0
        from .. import backend as _backend_mod

        be = _backend_mod._backend       # (plain bookkeeping: never instantiates the backend / loads the library)
        if be is not None and getattr(getattr(be, "_hip", None), "_one_launch_used", False) and self.device.type == "cuda":
            # the opt-in single-launch decode step: a sequence boundary is where its arrival tickets get audited
            be._hip.check_decode_tickets(self.device)

    def evict_last(self, num_tokens):
        g = self._decode_graph
        before = (tuple(self.kv_seq_len_list), tuple(self.streaming_kv_seq_len_list)) if g is not None else None
        for i in range(self.num_layers):
            self.kv_seq_len_list[i] = max(0, self.kv_seq_len_list[i] - num_tokens)
            self.streaming_kv_seq_len_list[i] = max(0, self.streaming_kv_seq_len_list[i] - num_tokens)
        if g is not None:       # an automatically captured decode step: its device-side counters follow with one launch
            g.host_evicted(before, num_tokens)

    # ---- a prefill chunk processed in row blocks (duo_static_attention_row_block) ------------------
    def begin_chunk(self, layer_idx, chunk_len):
        """Start a chunk of ``chunk_len`` rows on this layer: remember the cache length at its start and make
        sure the layer's staging buffer for the chunk's streaming K/V rows is large enough."""
        if not hasattr(self, "_chunk_state"):
            self._chunk_state = {}
        st = self._chunk_state.get(layer_idx)
        ns = self.num_streaming_kv_head_list[layer_idx]
        if st is None or st["stage_k"].shape[1] < chunk_len:
            st = {"stage_k": self._alloc(chunk_len, ns), "stage_v": self._alloc(chunk_len, ns)}
            self._chunk_state[layer_idx] = st
        st.update(past=self.kv_seq_len_list[layer_idx], chunk_len=chunk_len, next_row=0, open=True)
        return st

    def chunk_state(self, layer_idx):
        st = getattr(self, "_chunk_state", {}).get(layer_idx)
        if st is None or not st.get("open"):
            raise ValueError("no chunk in progress on this layer: the first row block must start at row 0")
        return st

    def end_chunk(self, layer_idx):
        self._chunk_state[layer_idx]["open"] = False

    # ---- device-side counters (a decode step that can be captured in a HIP graph) --------
    # The reference keeps the lengths as Python ints only (static_kv_cache.py:44-45), so every launch
    # bakes them in.  With the device state enabled each layer also has {full_len, str_len, pos, pad}
    # int32 in HBM; while ``use_device_state`` is set the fused decode step reads its lengths from there
    # (duo_decode_layer_dev_bf16), and ``device_state_add`` advances / rewinds all layers in one launch.
    # The Python ints stay the host's view (planning, capacity checks, every non-captured path).
    device_state = None
    use_device_state = False
    _decode_graph = None        # the automatically captured decode step of this cache, if any (duo_attn/graph.py)
    _device_counters = None     # the host counters the device copy is known to equal (None: unknown / never uploaded)

    def enable_device_state(self):
        if self.device_state is None:
            self.device_state = torch.zeros(self.num_layers, 4, dtype=torch.int32, device=self.device)
        self.sync_device_state()
        return self.device_state

    def sync_device_state(self):
        """host counters -> device (one small H2D copy; not capturable, call it outside a graph)"""
        rows = [[self.kv_seq_len_list[i], self.streaming_kv_seq_len_list[i], self.kv_seq_len_list[i], 0]
                for i in range(self.num_layers)]
        self.device_state.copy_(torch.tensor(rows, dtype=torch.int32), non_blocking=False)
        self._device_counters = (tuple(self.kv_seq_len_list), tuple(self.streaming_kv_seq_len_list))

    def device_state_add(self, d_full, d_str, d_pos):
        from ..backend import get_backend

        get_backend().decode_state_add(self.device_state, d_full, d_str, d_pos, self.sink_size + self.recent_size)

    @property
    def memory_usage(self):
        total = 0
        for lists in (self.full_key_states_list, self.full_value_states_list,
                      self.streaming_key_states_list, self.streaming_value_states_list):
            for t in lists:
                total += t.element_size() * t.numel()
        return total


# =============================================================================
# Static-cache forwards (reference static_kv_cache.py:318-567; llama == mistral)
# =============================================================================
def duo_attn_static_kv_cache_for_causal_lm_forward(
    self,
    input_ids: torch.LongTensor = None,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_values: Optional[DuoAttentionStaticKVCache] = None,
    inputs_embeds: Optional[torch.FloatTensor] = None,
    labels: Optional[torch.LongTensor] = None,
    use_cache: Optional[bool] = None,
    output_attentions: Optional[bool] = None,
    output_hidden_states: Optional[bool] = None,
    return_dict: Optional[bool] = None,
    **kwargs,
):
    # The reference's decode loop (eval/efficiency/benchmark_static.py:96-105) calls this once per token from Python; the
    # step is ~230 launches.  With DUO_AUTO_DECODE_GRAPH=1 (opt-in: worth it on a slow host or at short contexts), a call
    # that is exactly that loop's — one token, implicit positions, a static cache on this GPU, no gradients — is served by
    # a HIP graph captured on the way (duo_attn/graph.py: auto_decode_step); every other call, and the default, runs the
    # body below as it is.
    from ..graph import auto_decode_eligible, auto_decode_step

    if auto_decode_eligible(self, input_ids, position_ids, past_key_values, inputs_embeds, labels, kwargs):
        eager = lambda tok: duo_attn_static_kv_cache_for_causal_lm_forward(
            self, input_ids=tok, past_key_values=past_key_values, use_cache=use_cache, _duo_no_auto_graph=True)
        logits = auto_decode_step(self, eager, input_ids, past_key_values)
        if logits is not None:
            return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=past_key_values)
    kwargs.pop("_duo_no_auto_graph", None)
    outputs = self.model(
        input_ids=input_ids,
        attention_mask=attention_mask,
        position_ids=position_ids,
        past_key_values=past_key_values,
        inputs_embeds=inputs_embeds,
        use_cache=use_cache,
    )
    hidden_states = outputs.last_hidden_state
    pp = getattr(self, "_duo_pp", None)
    if pp is not None and pp.pipe.world_size > 1:
        # layer pipeline (one process per GPU): the logits exist on the last stage.  A decode step (S == 1) hands
        # them to every rank — the caller's argmax must agree everywhere; prefill chunks do not (that would
        # serialise the chunk pipeline), unless asked with sync_logits=True.
        logits = self.lm_head(hidden_states[:, -1:, :]) if pp.is_last else None
        S = (input_ids if input_ids is not None else inputs_embeds).shape[1]
        if S == 1 or kwargs.get("sync_logits", False):
            B = (input_ids if input_ids is not None else inputs_embeds).shape[0]
            dtype = next(self.model.layers[0].parameters()).dtype
            logits = pp.broadcast_from_last(logits, (B, 1, self.config.vocab_size), dtype)
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=outputs.past_key_values)
    if self.training:
        logits = self.lm_head(hidden_states).float()
    else:
        # eval: logits for the last position only (reference :360-364)
        logits = self.lm_head(hidden_states[:, -1:, :])
    loss = None
    if labels is not None:
        shift_logits = logits[..., :-1, :].contiguous().view(-1, self.config.vocab_size)
        shift_labels = labels[..., 1:].contiguous().view(-1).to(shift_logits.device)
        loss = torch.nn.functional.cross_entropy(shift_logits, shift_labels)
    return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=outputs.past_key_values)


def duo_attn_static_kv_cache_model_forward(
    self,
    input_ids: torch.LongTensor = None,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_values: Optional[DuoAttentionStaticKVCache] = None,
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

    past_len = past_key_values.kv_seq_len if past_key_values is not None else 0
    # The RoPE kernel takes the first position by value (the reference hands
    # position_ids[:, 0] to flashinfer, llama.py:350-352).  With implicit positions
    # that is just the cache length: no device read-back on the hot path.
    if position_ids is None:
        device = input_ids.device if input_ids is not None else inputs_embeds.device
        position_ids = torch.arange(past_len, seq_length + past_len, dtype=torch.long, device=device)
        position_ids = position_ids.unsqueeze(0).view(-1, seq_length)
        pos0 = past_len
    else:
        position_ids = position_ids.view(-1, seq_length).long()
        from ._duo import first_positions

        pos0 = first_positions(position_ids)     # per batch row when the rows differ (left-padded batches)

    pp = getattr(self, "_duo_pp", None)
    if pp is not None and not pp.is_first:
        # layer pipeline: this stage's input is the previous stage's hidden state (RCCL point-to-point)
        bsz = (input_ids if input_ids is not None else inputs_embeds).shape[0]
        hidden_states = pp.recv_hidden((bsz, seq_length, self.config.hidden_size),
                                       next(self.layers[0].parameters()).dtype)
    else:
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids.to(self.embed_tokens.weight.device))
        hidden_states = inputs_embeds
    for idx, decoder_layer in enumerate(self.layers):       # (sharded model: the layers this rank kept)
        hidden_states = decoder_layer(
            hidden_states,
            position_ids=position_ids,
            kv_cache=past_key_values,
            layer_idx=idx,
            use_cache=use_cache,
            pos0=pos0,
        )[0]
    if pp is not None and not pp.is_last:
        pp.send_hidden(hidden_states)       # asynchronous: the next chunk starts here while this one moves on
    else:
        hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=past_key_values)


def duo_attn_static_kv_cache_decoder_layer_forward(
    self,
    hidden_states: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    kv_cache: Optional[DuoAttentionStaticKVCache] = None,
    layer_idx: int = None,
    output_attentions: Optional[bool] = False,
    use_cache: Optional[bool] = False,
    **kwargs,
):
    if hidden_states.shape[1] == 1 and "row_block" not in kwargs:
        # decode step: the token-row linears either side of the attention op fused around it (six launches per layer)
        from ._duo import duo_decode_layer_fused, fused_decode_layer_ok

        if fused_decode_layer_ok(self, hidden_states, kv_cache, layer_idx):
            return (duo_decode_layer_fused(self, hidden_states, kv_cache, layer_idx, kwargs.get("pos0"), position_ids),)
    residual = hidden_states
    hidden_states = self.input_layernorm(hidden_states)
    hidden_states, _ = self.self_attn(
        hidden_states=hidden_states,
        position_ids=position_ids,
        kv_cache=kv_cache,
        layer_idx=layer_idx,
        use_cache=use_cache,
        **kwargs,
    )
    hidden_states = residual + hidden_states
    residual = hidden_states
    hidden_states = self.post_attention_layernorm(hidden_states)
    hidden_states = _mlp_forward(self.mlp, hidden_states)
    hidden_states = residual + hidden_states
    return (hidden_states,)


def _mlp_forward(mlp, x):
    """HF LlamaMLP / MistralMLP: ``down_proj(act_fn(gate_proj(x)) * up_proj(x))`` with the activation product as ONE
    elementwise pass (duo_silu_mul_bf16) instead of a SiLU kernel and a multiply — same roundings; anything that is not a
    SiLU-gated bf16 MLP on the GPU (or a backend without the kernel, or an MLP / activation module carrying a forward hook —
    the bypass would not fire it) runs the module as it is"""
    from ..backend import get_backend
    from ._duo import modules_hooked

    be = get_backend()
    if (hasattr(be, "silu_mul") and x.is_cuda and x.dtype == torch.bfloat16
            and type(getattr(mlp, "act_fn", None)).__name__ in ("SiLUActivation", "SiLU")
            and all(hasattr(mlp, n) for n in ("gate_proj", "up_proj", "down_proj"))
            and not modules_hooked((mlp, mlp.act_fn))):
        g, u = mlp.gate_proj(x), mlp.up_proj(x)
        # (the kernel moves 16-byte pieces: inner dimension a multiple of 8, unit inner stride, 16-byte aligned rows —
        #  anything else takes torch's two kernels, same roundings)
        if (g.shape == u.shape and g.shape[-1] % 8 == 0 and g.dtype == u.dtype == torch.bfloat16 and g.stride(-1) == 1
                and u.stride(-1) == 1 and g.data_ptr() % 16 == 0 and u.data_ptr() % 16 == 0):
            return mlp.down_proj(be.silu_mul(g, u))
        return mlp.down_proj(mlp.act_fn(g) * u)
    return mlp(x)


def enable_duo_attention_static_kv_cache(model):
    """Rebind LM / model / layer forwards (reference :554-567)."""
    model.model.forward = types.MethodType(duo_attn_static_kv_cache_model_forward, model.model)
    for layer in model.model.layers:
        layer.forward = types.MethodType(duo_attn_static_kv_cache_decoder_layer_forward, layer)
    model.forward = types.MethodType(duo_attn_static_kv_cache_for_causal_lm_forward, model)


# family-named aliases, as imported by reference call sites
enable_duo_attention_static_kv_cache_for_llama = enable_duo_attention_static_kv_cache
enable_duo_attention_static_kv_cache_for_mistral = enable_duo_attention_static_kv_cache
duo_attn_static_kv_cache_llama_for_causal_lm_forward = duo_attn_static_kv_cache_for_causal_lm_forward
duo_attn_static_kv_cache_llama_model_forward = duo_attn_static_kv_cache_model_forward
duo_attn_static_kv_cache_llama_decoder_layer_forward = duo_attn_static_kv_cache_decoder_layer_forward
duo_attn_static_kv_cache_mistral_for_causal_lm_forward = duo_attn_static_kv_cache_for_causal_lm_forward
duo_attn_static_kv_cache_mistral_model_forward = duo_attn_static_kv_cache_model_forward
duo_attn_static_kv_cache_mistral_decoder_layer_forward = duo_attn_static_kv_cache_decoder_layer_forward
