"""Device backend of the DuoAttention hot path.

The product ships exactly ONE backend: :class:`HipBackend`, a thin adapter from
torch tensor views to the C-ABI library (``_hip.py``).  It refuses CPU tensors
and raises if the library is missing — there is no CPU fallback.

``_set_backend_for_testing`` exists so that the CPU test-suite can drive the
*host plumbing* (patch API, cache bookkeeping, weight reordering, pipeline
schedule) with the oracle from ``oracle/`` plugged in.  Nothing in this package
imports ``oracle``.

Tensor conventions (B = 1 slice already taken by the caller):
    q, out          [S, Hq, D]   bf16 views, D contiguous
    K/V segment     ([T, h, D], [T, h, D]) views or None
    class desc      (n_kv_heads, q_head_offset, segA, segB)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

Seg = Optional[Tuple[torch.Tensor, torch.Tensor]]
ClassDesc = Optional[Tuple[int, int, Seg, Seg]]


class HipBackend:
    name = "hip"

    def __init__(self):
        from . import _hip

        self._hip = _hip
        _hip.load_library()  # fail loudly, now

    # -- RoPE (flashinfer.rope.apply_rope_inplace; reference flashinfer_utils.py:29-59)
    def rope_inplace(self, q, k, pos0: int, rope_scale: float, rope_theta: float):
        self._hip.rope_inplace(q, k, pos0, rope_scale, rope_theta)

    # -- pool[dst_row0 : dst_row0+S] = src   (reference static_kv_cache.py:109-125)
    def kv_append(self, k_src, v_src, k_pool, v_pool, dst_row0: int):
        self._hip.kv_append(k_src, v_src, k_pool, v_pool, dst_row0)

    # -- sink+recent update in place (reference static_kv_cache.py:127-167); returns new length
    def stream_compress(self, k_pool, v_pool, k_new, v_new, cur_len: int, sink: int, recent: int) -> int:
        return self._hip.stream_compress(k_pool, v_pool, k_new, v_new, cur_len, sink, recent)

    def _cls(self, desc: ClassDesc, dtype=torch.bfloat16):
        if desc is None or desc[0] <= 0:
            return None
        n_kv, q_off, a, b = desc
        h = self._hip
        return h.make_class(n_kv, q_off, h.make_seg(*(a or (None, None)), dtype=dtype),
                            h.make_seg(*(b or (None, None)), dtype=dtype))

    # -- the flash_attn_func calls of reference llama.py:364-421
    def attention(self, q, out, group: int, full: ClassDesc, stream: ClassDesc, scale: float):
        fc, sc = self._cls(full, q.dtype), self._cls(stream, q.dtype)
        if q.shape[0] == 1 and q.dtype == torch.bfloat16:
            self._hip.attn_decode(q[0], out[0], group, fc, sc, scale)
        else:
            # (fp16 — the INT4 demo's dequantised scratch — has no split-KV scan of its own: its decode steps read the packed
            #  pools (attn_decode_int4), and a ONE-TOKEN chunk over fp16 rows, e.g. a one-token prompt, is a one-row query
            #  block of the MFMA kernel)
            self._hip.attn_prefill(q, out, group, fc, sc, scale)

    # -- batched forms: the batch row is a grid dimension of the same kernels (one launch for all rows).  Views carry a
    #    leading batch dimension: q/out [B, S, Hq, D], segments [B, T, h, D]
    def attention_batched(self, q, out, group: int, full: ClassDesc, stream: ClassDesc, scale: float):
        fc, sc = self._cls(full, q.dtype), self._cls(stream, q.dtype)
        self._hip.attention_batched(q, out, group, fc, sc, scale)

    def rope_inplace_batched(self, q, k, pos0, rope_scale: float, rope_theta: float):
        self._hip.rope_inplace_batched(q, k, pos0, rope_scale, rope_theta)

    def kv_append_batched(self, k_src, v_src, k_pool, v_pool, dst_row0: int):
        self._hip.kv_append_batched(k_src, v_src, k_pool, v_pool, dst_row0)

    def stream_compress_batched(self, k_pool, v_pool, k_new, v_new, cur_len: int, sink: int, recent: int) -> int:
        return self._hip.stream_compress_batched(k_pool, v_pool, k_new, v_new, cur_len, sink, recent)

    def decode_layer_batched(self, *args, **kw) -> int:
        return self._hip.decode_layer_batched(*args, **kw)

    # -- a whole decode step of one layer (reference llama.py:332-425, q_len == 1): scan + epilogue launch (or one launch)
    def decode_layer(self, q, k, v, out, n_full, full_k, full_v, full_len, str_k, str_v, str_len, sink,
                     recent, pos, rope_scale, rope_theta, scale) -> int:
        return self._hip.decode_layer(q, k, v, out, n_full, full_k, full_v, full_len, str_k, str_v, str_len,
                                      sink, recent, pos, rope_scale, rope_theta, scale)

    def decode_layer_dev(self, *args, **kw) -> None:
        return self._hip.decode_layer_dev(*args, **kw)

    def decode_state_add(self, dev_states, d_full, d_str, d_pos, str_cap) -> None:
        return self._hip.decode_state_add(dev_states, d_full, d_str, d_pos, str_cap)

    # -- flashinfer.norm.rmsnorm (reference flashinfer_utils.py:9-16)
    def rmsnorm(self, x, weight, eps: float):
        return self._hip.rmsnorm(x, weight, eps)

    # -- the torch.nn.Linear / RMSNorm / SiLU*mul / residual-add modules either side of the attention op at q_len == 1
    #    (reference llama.py:332-340, :430-432; static_kv_cache.py:482-537): weight rows streamed once, one launch each
    def silu_mul(self, gate, up):
        return self._hip.silu_mul(gate, up)

    def token_linear_fits(self, n_rows: int, n_in: int) -> bool:
        return self._hip.token_linear_fits(n_rows, n_in)

    def token_linear(self, x, blocks, norm=None, x2=None, residual=None, norm_hf=False):
        return self._hip.token_linear(x, blocks, norm=norm, x2=x2, residual=residual, norm_hf=norm_hf)

    # -- the tuple-cache forward's data movement at q_len == 1 (reference llama.py:177-184, :202-223, :273-301): HF rotary in
    #    place, retrieval rows appended to the arena, the new (truncated) streaming cache written out of place — one launch
    def tuple_decode_prep(self, q, k, v, cos_row, sin_row, n_full, arena, full_len, str_src, sink, recent):
        return self._hip.tuple_decode_prep(q, k, v, cos_row, sin_row, n_full, arena, full_len, str_src, sink, recent)

    # -- the tuple path on prefill chunks: HF rotary (llama.py:177-184) and HF's *RMSNorm.forward as single passes
    def rope_hf_inplace(self, q, k, cos, sin):
        self._hip.rope_hf_inplace(q, k, cos, sin)

    def rmsnorm_hf(self, x, weight, eps: float):
        return self._hip.rmsnorm_hf(x, weight, eps)

    # -- ... and its two flash_attn_func calls (llama.py:225-262) over the tuple format, segments described by strides
    #    (same kernel as `attention` at q_len == 1; no tensor views on the host)
    def tuple_decode_attention(self, q, out, group, n_full, arena, full_len, str_src, k, v, scale):
        self._hip.attn_decode_tuple(q, out, group, n_full, arena, full_len, str_src, k, v, scale)

    def decode_layer_batched_dev(self, *args, **kw) -> None:
        return self._hip.decode_layer_batched_dev(*args, **kw)


_backend = None


def get_backend():
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def _set_backend_for_testing(backend):
    """Test hook: plug a checker backend (the CPU oracle) in; ``None`` restores HIP."""
    global _backend
    _backend = backend
