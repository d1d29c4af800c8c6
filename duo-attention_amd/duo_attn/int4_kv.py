"""INT4 KV pools for the static dual cache (reference ``demo/int4_kv.py``, BASELINE config 5).

``DuoAttentionStaticINT4KVCache`` keeps the reference's constructor, counters and methods
(:115-492): ``put`` quantises the new K/V rows and appends them to the retrieval and streaming
pools, ``get`` dequantises whole pools into fp16 scratch (kept because callers of the reference use
it — the decode path below does not), ``compress`` keeps sink + recent rows of the streaming pool.
Differences that do not change the interface: pools are head-major in HBM (one contiguous stream per
kv head) behind token-major views; the fp16 (scale, zero) pair of a row is stored adjacently; rows are
quantised straight into the pool (no staging buffers); and ``decode_attention`` runs the attention of a
single-token step directly on the packed pools (``duo_attn_decode_int4_f16``), which is what turns the
reference's per-step "dequantise 100 % of the cache, write it, read it back" into one read of 136 B per
row.  fp16, like the reference's QServe model (``demo/w8a8kv4_llama.py``).
"""
from __future__ import annotations

import torch

from . import _hip

GROUP_SIZE = 128


class QuantizedCache:
    """reference int4_kv.py:4-43 — same attribute names; ``scale`` / ``zero_point`` are views of one
    interleaved (scale, zero) tensor."""

    def __init__(self, batch_size, max_size, num_kv_heads, head_dim, device, group_size):
        assert group_size == head_dim == GROUP_SIZE, "one quantisation group per row (reference :140)"
        self.batch_size, self.max_size = batch_size, max_size
        self.num_kv_heads, self.head_dim, self.device, self.group_size = num_kv_heads, head_dim, device, group_size
        self.num_groups = 1
        # (a head's rows are padded to a multiple of 4: its (scale, zero) words then start on a 16-byte boundary, which the
        #  folded decode kernel's 16-byte fetches of four keys' pairs want; the logical shapes are the reference's)
        rows = (max_size + 3) // 4 * 4
        q = torch.zeros(batch_size, num_kv_heads, rows, head_dim // 2, device=device, dtype=torch.uint8)
        sz = torch.zeros(batch_size, num_kv_heads, rows, 2, device=device, dtype=torch.float16)
        self.quantized_data = q.permute(0, 2, 1, 3)[:, :max_size]      # logical [B, T, h, 64]
        self.scale_zero = sz.permute(0, 2, 1, 3)[:, :max_size]         # logical [B, T, h, 2]
        self.scale = self.scale_zero[..., 0:1]
        self.zero_point = self.scale_zero[..., 1:2]


class DuoAttentionStaticINT4KVCache:
    def __init__(self, model, full_attention_heads, batch_size, max_size, sink_size, recent_size,
                 prefilling_chunk_size, fused_dequant: bool = False, folded_decode: bool = False):
        """``fused_dequant``: dequantise as ONE fma(q, s, z) instead of the source's hmul-then-hadd (two roundings, the
        default here).  Which of the two the reference's own build computes depends on whether its compiler contracts
        ``__hadd(__hmul(f, s), z)``; both forms are pinned against builds of the reference's kernel.
        ``folded_decode`` (opt-in): the single-token attention applies scale and zero to the score tile and to P instead of
        dequantising every element, in every 32-key tile whose rows are tame (scales < 1, zero points in (-8, 8)): the
        attention over ``n s + z`` WITHOUT either form's fp16 roundings (``duo_int4_decode_fold_kernel``); tiles with outlier
        rows dequantise element by element in the ``fused_dequant`` form.  9-15 % faster per token at multi-million-token
        contexts; its outputs differ from the reference's by the reference's own value rounding — inside the bar that
        budgets one fp16 ulp per dequantised value, outside the strict decode bar the default meets (DESIGN §3,
        profiles/r4_int4_fold.md).  Default ``False``: every tile dequantises in registers, bit for bit the values ``get()``
        would have written to scratch."""
        self.fused_dequant = bool(fused_dequant)
        self.folded_decode = bool(folded_decode)
        self.batch_size, self.max_size = batch_size, max_size
        self.sink_size, self.recent_size = sink_size, recent_size
        self.prefilling_chunk_size = prefilling_chunk_size
        self.device = next(model.parameters()).device
        self.dtype = next(model.parameters()).dtype
        cfg = model.config
        # a model sharded over pipeline stages (duo_attn.pipeline.shard_model_for_pp): this rank owns the pools of the
        # layers it kept, indexed by LOCAL layer number like the patched forwards do; `full_attention_heads` may be the
        # whole model's pattern or already the stage's rows (same rule as DuoAttentionStaticKVCache)
        pp = getattr(model, "_duo_pp", None)
        if pp is not None:
            full_attention_heads = pp.local_rows(full_attention_heads)
        full_attention_heads = list(full_attention_heads)
        self.num_layers = len(full_attention_heads) if pp is not None else cfg.num_hidden_layers
        if len(full_attention_heads) != self.num_layers:
            raise ValueError(f"{len(full_attention_heads)} head-pattern rows for {self.num_layers} layers")
        self.num_heads = cfg.num_attention_heads
        self.num_kv_heads = cfg.num_key_value_heads
        self.num_kv_groups = self.num_heads // self.num_kv_heads
        # (an explicit config.head_dim wins over hidden_size // num_heads — they differ in some checkpoints
        # and in a tensor-parallel shard, where the head count is per rank)
        self.head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // self.num_heads
        self.group_size = GROUP_SIZE
        self.num_full_kv_head_list, self.num_streaming_kv_head_list = [], []
        self.streaming_key_caches, self.streaming_value_caches = [], []
        self.full_key_caches, self.full_value_caches = [], []
        max_nf = max_ns = 0
        cap_s = sink_size + recent_size + prefilling_chunk_size
        for heads in full_attention_heads:
            nf = int((torch.as_tensor(heads) > 0.5).sum().item())
            ns = self.num_kv_heads - nf
            max_nf, max_ns = max(max_nf, nf), max(max_ns, ns)
            self.num_full_kv_head_list.append(nf)
            self.num_streaming_kv_head_list.append(ns)
            mk = lambda rows, h: QuantizedCache(batch_size, rows, h, self.head_dim, self.device, GROUP_SIZE)
            self.streaming_key_caches.append(mk(cap_s, ns))
            self.streaming_value_caches.append(mk(cap_s, ns))
            self.full_key_caches.append(mk(max_size, nf))
            self.full_value_caches.append(mk(max_size, nf))
        self.kv_seq_len_list = [0] * self.num_layers
        self.streaming_kv_seq_len_list = [0] * self.num_layers
        # fp16 scratch of get() — allocated lazily: the fused decode path never needs it
        self._scratch = {}
        self._max_nf, self._max_ns, self._cap_s = max_nf, max_ns, cap_s

    @property
    def streaming_kv_seq_len(self):
        return self.streaming_kv_seq_len_list[-1]

    @property
    def kv_seq_len(self):
        return self.kv_seq_len_list[-1]

    def _buf(self, name, numel):
        b = self._scratch.get(name)
        if b is None or b.numel() < numel:
            b = torch.empty(numel, device=self.device, dtype=torch.float16)
            self._scratch[name] = b
        return b

    # ---------------------------------------------------------------- put (reference :261-371)
    def put(self, layer_idx, key_states, value_states, dequantize=True):
        nf = self.num_full_kv_head_list[layer_idx]
        incoming = key_states.shape[1]
        cur, cur_s = self.kv_seq_len_list[layer_idx], self.streaming_kv_seq_len_list[layer_idx]
        if incoming + cur > self.max_size:
            raise ValueError(
                f"Trying to put {incoming} KVs into a cache with max size {self.max_size}, current size: {cur}."
            )
        # (a layer without streaming heads never compresses: its counter just grows, reference :444)
        if self.num_streaming_kv_head_list[layer_idx] > 0 and incoming + cur_s > self._cap_s:
            raise ValueError(
                f"Trying to put {incoming} KVs into a streaming cache of {self._cap_s} rows, current size: {cur_s}."
            )
        for src, fc, sc in ((key_states, self.full_key_caches, self.streaming_key_caches),
                            (value_states, self.full_value_caches, self.streaming_value_caches)):
            # every batch row of a head class in ONE launch (batch row = a grid dimension)
            _hip.int4_quantize_batched(src[:, :, :nf], fc[layer_idx].quantized_data, fc[layer_idx].scale_zero, cur)
            _hip.int4_quantize_batched(src[:, :, nf:], sc[layer_idx].quantized_data, sc[layer_idx].scale_zero, cur_s)
        self.kv_seq_len_list[layer_idx] += incoming
        self.streaming_kv_seq_len_list[layer_idx] += incoming
        return self.get(layer_idx) if dequantize else None

    # ---------------------------------------------------------------- get (reference :373-436)
    def get(self, layer_idx):
        n, m = self.kv_seq_len_list[layer_idx], self.streaming_kv_seq_len_list[layer_idx]
        nf, ns = self.num_full_kv_head_list[layer_idx], self.num_streaming_kv_head_list[layer_idx]
        B = self.batch_size
        empty = torch.empty(0, device=self.device, dtype=torch.float16)

        def deq(cache, rows, heads, name):
            if heads == 0:
                return empty
            out = self._buf(name, B * rows * heads * self.head_dim)
            return _hip.int4_dequantize_batched(cache.quantized_data, cache.scale_zero, rows, out, fused=self.fused_dequant)

        return (deq(self.full_key_caches[layer_idx], n, nf, "fk"), deq(self.full_value_caches[layer_idx], n, nf, "fv"),
                deq(self.streaming_key_caches[layer_idx], m, ns, "sk"), deq(self.streaming_value_caches[layer_idx], m, ns, "sv"))

    # ---------------------------------------------------------------- compress (reference :438-492)
    def compress(self, layer_idx):
        m = self.streaming_kv_seq_len_list[layer_idx]
        if m <= self.recent_size + self.sink_size:
            return
        kc, vc = self.streaming_key_caches[layer_idx], self.streaming_value_caches[layer_idx]
        if self.num_streaming_kv_head_list[layer_idx] > 0:
            _hip.int4_stream_compress_batched(kc.quantized_data, kc.scale_zero, vc.quantized_data, vc.scale_zero, m,
                                              self.sink_size, self.recent_size)
            # (the reference only moves the counter when the layer has streaming heads, :444-492)
            self.streaming_kv_seq_len_list[layer_idx] = self.recent_size + self.sink_size

    def clear(self):
        for i in range(self.num_layers):
            self.kv_seq_len_list[i] = 0
            self.streaming_kv_seq_len_list[i] = 0

    def evict_last(self, num_tokens):
        for i in range(self.num_layers):
            self.kv_seq_len_list[i] = max(0, self.kv_seq_len_list[i] - num_tokens)
            self.streaming_kv_seq_len_list[i] = max(0, self.streaming_kv_seq_len_list[i] - num_tokens)

    @property
    def memory_usage(self):
        total = 0
        for caches in (self.full_key_caches, self.full_value_caches, self.streaming_key_caches,
                       self.streaming_value_caches):
            for c in caches:
                total += c.quantized_data.numel() + 2 * c.scale_zero.numel()
        return total

    # ---------------------------------------------------------------- chunked-prefill attention
    def prefill_attention(self, layer_idx, query_states, key_states, value_states, scale=None):
        """The attention of reference demo/w8a8kv4_llama.py:226-274 for q_len > 1, called AFTER ``put()``:
        * first chunk (the cache held nothing before this put): every head causal over the chunk's own,
          un-quantised K/V (:226-234);
        * later chunks: retrieval heads over the dequantised full pool, streaming heads over the
          dequantised streaming pool (both already contain the chunk's quantised rows), bottom-right
          causal (:236-274).
        query_states [B, S, Hq, 128], key/value_states [B, S, Hkv, 128], fp16, after RoPE.  The pools are
        dequantised into this object's fp16 scratch (as the reference's ``get`` does) and read by the fp16
        MFMA kernel; ``compress()`` is the caller's next step, as in the reference."""
        from .backend import get_backend

        be = get_backend()
        B, S = query_states.shape[0], query_states.shape[1]
        nf, ns = self.num_full_kv_head_list[layer_idx], self.num_streaming_kv_head_list[layer_idx]
        n, m = self.kv_seq_len_list[layer_idx], self.streaming_kv_seq_len_list[layer_idx]
        G = self.num_kv_groups
        scale = self.head_dim ** -0.5 if scale is None else scale
        out = torch.empty_like(query_states)
        batched = B > 1 and hasattr(be, "attention_batched")      # fp16 prefill: one launch for all batch rows
        if n == S:      # first chunk
            if batched:
                be.attention_batched(query_states, out, G, (nf + ns, 0, None, (key_states, value_states)), None, scale)
            for b in range(0 if batched else B):
                be.attention(query_states[b], out[b], G, (nf + ns, 0, None, (key_states[b], value_states[b])), None, scale)
            return out
        fk, fv, sk, sv = self.get(layer_idx)
        if batched:
            full = (nf, 0, (fk[:, :n - S], fv[:, :n - S]), (fk[:, n - S:n], fv[:, n - S:n])) if nf else None
            stream = (ns, nf * G, (sk[:, :m - S], sv[:, :m - S]), (sk[:, m - S:m], sv[:, m - S:m])) if ns else None
            be.attention_batched(query_states, out, G, full, stream, scale)
            return out
        for b in range(B):
            full = (nf, 0, (fk[b, :n - S], fv[b, :n - S]), (fk[b, n - S:n], fv[b, n - S:n])) if nf else None
            stream = (ns, nf * G, (sk[b, :m - S], sv[b, :m - S]), (sk[b, m - S:m], sv[b, m - S:m])) if ns else None
            be.attention(query_states[b], out[b], G, full, stream, scale)
        return out

    # ---------------------------------------------------------------- fused decode attention
    def decode_attention(self, layer_idx, query_states, scale=None):
        """query_states [B, 1, Hq, 128] fp16 (after RoPE, after put()): attention of the decode branch
        of reference demo/w8a8kv4_llama.py:240-274 over the dequantised pools, computed on the packed
        pools.  Returns [B, 1, Hq, 128] fp16."""
        nf, ns = self.num_full_kv_head_list[layer_idx], self.num_streaming_kv_head_list[layer_idx]
        n, m = self.kv_seq_len_list[layer_idx], self.streaming_kv_seq_len_list[layer_idx]
        G = self.num_kv_groups
        out = torch.empty_like(query_states)
        scale = self.head_dim ** -0.5 if scale is None else scale
        fk, fv = self.full_key_caches[layer_idx], self.full_value_caches[layer_idx]
        sk, sv = self.streaming_key_caches[layer_idx], self.streaming_value_caches[layer_idx]
        full = _hip.make_int4_pool(fk.quantized_data, fk.scale_zero, fv.quantized_data, fv.scale_zero, n, 0) if nf else None
        stream = _hip.make_int4_pool(sk.quantized_data, sk.scale_zero, sv.quantized_data, sv.scale_zero, m, nf * G) if ns else None
        _hip.attn_decode_int4_batched(query_states[:, 0], out[:, 0], G, full, stream, scale,
                                      fused=(2 + int(self.fused_dequant)) if self.folded_decode else self.fused_dequant)
        return out
