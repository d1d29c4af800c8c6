"""ctypes binding of the C-ABI library ``libduoattn_hip.so`` (include/duo_attn_hip.h).

This is the only place the Python host touches native code.  Tensors cross the
boundary as raw device pointers + element strides + the current HIP stream;
PyTorch is used for device memory and streams only.  There is NO CPU fallback:
a missing library, a non-GPU tensor or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p
from typing import Optional

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_PKG_DIR, "..", "lib", "libduoattn_hip.so"))
ABI_VERSION = 6
HEAD_DIM = 128


class DuoHipError(RuntimeError):
    pass


class KVSeg(Structure):
    """``duo_kv_seg``"""

    _fields_ = [
        ("k", c_void_p),
        ("v", c_void_p),
        ("token_stride", c_int64),
        ("head_stride", c_int64),
        ("len", c_int32),
        ("_pad", c_int32),
        ("batch_stride", c_int64),
    ]


class HeadClass(Structure):
    """``duo_head_class``"""

    _fields_ = [
        ("n_kv_heads", c_int32),
        ("q_head_offset", c_int32),
        ("segA", KVSeg),
        ("segB", KVSeg),
    ]


class Int4Pool(Structure):
    """``duo_int4_pool``"""

    _fields_ = [
        ("k_q", c_void_p), ("v_q", c_void_p), ("k_sz", c_void_p), ("v_sz", c_void_p),
        ("token_stride_rows", c_int64), ("head_stride_rows", c_int64),
        ("len", c_int32), ("n_kv_heads", c_int32), ("q_head_offset", c_int32), ("_pad", c_int32),
        ("batch_stride_rows", c_int64),
    ]


class DecodeLayerArgs(Structure):
    """``duo_decode_layer_args``"""

    _fields_ = [
        ("q", c_void_p), ("q_head_stride", c_int64), ("n_q_heads", c_int32), ("n_kv_heads", c_int32),
        ("k", c_void_p), ("v", c_void_p), ("kv_head_stride", c_int64),
        ("out", c_void_p), ("out_head_stride", c_int64),
        ("n_full", c_int32), ("head_dim", c_int32),
        ("full_k", c_void_p), ("full_v", c_void_p), ("full_token_stride", c_int64), ("full_head_stride", c_int64),
        ("full_len", c_int32), ("full_capacity", c_int32),
        ("str_k", c_void_p), ("str_v", c_void_p), ("str_token_stride", c_int64), ("str_head_stride", c_int64),
        ("str_len", c_int32), ("sink", c_int32), ("recent", c_int32), ("_pad", c_int32),
        ("pos", c_int64), ("rope_scale", c_float), ("rope_theta", c_float), ("scale", c_float), ("_pad2", c_float),
    ]


class DecodeBatch(Structure):
    """``duo_decode_batch``"""

    _fields_ = [
        ("n_batch", c_int32), ("_pad", c_int32),
        ("q_batch_stride", c_int64), ("kv_batch_stride", c_int64), ("out_batch_stride", c_int64),
        ("full_batch_stride", c_int64), ("str_batch_stride", c_int64),
        ("pos", POINTER(c_int64)),
    ]


_PREFILL_BATCHED = [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32,
                    POINTER(HeadClass), POINTER(HeadClass), c_float, c_int32, c_void_p, c_int64, c_void_p]
_ROPE_BATCHED = [c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32,
                 c_int32, POINTER(c_int64), c_float, c_float, c_int32, c_void_p]

class LinearSeg(Structure):
    """duo_linear_seg: one [n, n_in] bf16 weight block of a token-row linear (torch.nn.Linear.weight layout)."""
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("row_stride", c_int64), ("n", c_int32), ("reserved", c_int32)]


class TokenLinearArgs(Structure):
    """duo_token_linear_args (include/duo_attn_hip.h)."""
    _fields_ = [("x", c_void_p), ("x2", c_void_p), ("x_row_stride", c_int64), ("n_rows", c_int32), ("n_in", c_int32),
                ("seg", LinearSeg * 3), ("norm_weight", c_void_p), ("norm_eps", c_float), ("flags", c_int32),
                ("residual", c_void_p), ("residual_row_stride", c_int64), ("y", c_void_p), ("y_row_stride", c_int64)]


TOKEN_LINEAR_MAX_ROWS = 4      # DUO_TOKEN_LINEAR_MAX_ROWS
TOKEN_LINEAR_PAD = 2048        # DUO_TOKEN_LINEAR_PAD: token rows are staged in LDS padded to a multiple of this
LINEAR_NORM_HF = 1             # DUO_LINEAR_NORM_HF


class TupleDecodeArgs(Structure):
    """``duo_tuple_decode_args``"""

    _fields_ = [
        ("q", c_void_p), ("q_head_stride", c_int64), ("n_q_heads", c_int32), ("n_kv_heads", c_int32),
        ("k", c_void_p), ("v", c_void_p), ("kv_head_stride", c_int64),
        ("cos_row", c_void_p), ("sin_row", c_void_p),
        ("n_full", c_int32), ("head_dim", c_int32),
        ("full_k", c_void_p), ("full_v", c_void_p), ("full_token_stride", c_int64), ("full_head_stride", c_int64),
        ("full_len", c_int32), ("full_capacity", c_int32),
        ("str_k_src", c_void_p), ("str_v_src", c_void_p), ("src_token_stride", c_int64), ("src_head_stride", c_int64),
        ("str_k_dst", c_void_p), ("str_v_dst", c_void_p), ("dst_token_stride", c_int64), ("dst_head_stride", c_int64),
        ("str_len", c_int32), ("sink", c_int32), ("recent", c_int32), ("_pad", c_int32),
    ]


_SIGNATURES = {
    "duo_abi_version": (ctypes.c_int, []),
    "duo_target_arch": (c_char_p, []),
    "duo_error_string": (c_char_p, [ctypes.c_int]),
    "duo_set_debug_flags": (None, [c_uint32]),
    "duo_get_debug_flags": (c_uint32, []),
    "duo_debug_prefill_last_plan": (None, [POINTER(ctypes.c_double)]),
    "duo_debug_prefill_plan": (c_int32, [POINTER(c_int32), c_uint32, POINTER(c_int32), POINTER(ctypes.c_double),
                                         POINTER(c_int32), c_int32]),
    "duo_rope_inplace_bf16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int64,
         c_float, c_float, c_int32, c_void_p],
    ),
    "duo_rope_inplace_f16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int64,
         c_float, c_float, c_int32, c_void_p],
    ),
    "duo_kv_append_bf16": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_void_p],
    ),
    "duo_stream_compress_bf16": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), c_void_p],
    ),
    "duo_attn_decode_workspace_bytes": (c_int64, [c_int32, c_int32]),
    "duo_attn_decode_bf16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_int32, POINTER(HeadClass), POINTER(HeadClass), c_float,
         c_int32, c_void_p, c_int64, c_void_p],
    ),
    "duo_decode_layer_bf16": (
        ctypes.c_int, [POINTER(DecodeLayerArgs), POINTER(c_int32), c_void_p, c_int64, c_void_p],
    ),
    "duo_decode_layer_dev_bf16": (
        ctypes.c_int, [POINTER(DecodeLayerArgs), c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "duo_decode_step_bf16": (
        ctypes.c_int, [POINTER(DecodeLayerArgs), POINTER(c_int32), c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "duo_decode_state_add": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "duo_decode_plan_bucket": (c_int32, [c_int32]),
    "duo_decode_layer_batched_dev_bf16": (
        ctypes.c_int, [POINTER(DecodeLayerArgs), POINTER(DecodeBatch), c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "duo_tuple_decode_prep_bf16": (ctypes.c_int, [POINTER(TupleDecodeArgs), POINTER(c_int32), c_void_p]),
    "duo_rope_hf_inplace_bf16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int64,
         c_int32, c_void_p],
    ),
    "duo_rmsnorm_hf_bf16": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "duo_attn_prefill_bf16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, POINTER(HeadClass),
         POINTER(HeadClass), c_float, c_int32, c_void_p],
    ),
    "duo_attn_prefill_f16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, POINTER(HeadClass),
         POINTER(HeadClass), c_float, c_int32, c_void_p],
    ),
    "duo_attn_prefill_workspace_bytes": (c_int64, []),
    "duo_attn_prefill_batched_bf16": (ctypes.c_int, _PREFILL_BATCHED),
    "duo_attn_prefill_batched_f16": (ctypes.c_int, _PREFILL_BATCHED),
    "duo_attn_decode_batched_bf16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, POINTER(HeadClass),
         POINTER(HeadClass), c_float, c_int32, c_void_p, c_int64, c_void_p],
    ),
    "duo_decode_layer_batched_bf16": (
        ctypes.c_int, [POINTER(DecodeLayerArgs), POINTER(DecodeBatch), POINTER(c_int32), c_void_p, c_int64, c_void_p],
    ),
    "duo_rope_inplace_batched_bf16": (ctypes.c_int, _ROPE_BATCHED),
    "duo_rope_inplace_batched_f16": (ctypes.c_int, _ROPE_BATCHED),
    "duo_kv_append_batched_bf16": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_int32, c_void_p],
    ),
    "duo_stream_compress_batched_bf16": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), c_void_p],
    ),
    "duo_attn_prefill_ws_bf16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, POINTER(HeadClass),
         POINTER(HeadClass), c_float, c_int32, c_void_p, c_int64, c_void_p],
    ),
    "duo_attn_prefill_ws_f16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, POINTER(HeadClass),
         POINTER(HeadClass), c_float, c_int32, c_void_p, c_int64, c_void_p],
    ),
    "duo_rmsnorm_bf16": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "duo_token_linear_bf16": (ctypes.c_int, [POINTER(TokenLinearArgs), c_void_p]),
    "duo_silu_mul_bf16": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    "duo_int4_quantize": (
        ctypes.c_int,
        [c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
         c_int32, c_void_p],
    ),
    "duo_int4_dequantize_f16": (
        ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p],
    ),
    "duo_int4_stream_compress": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32,
         POINTER(c_int32), c_void_p],
    ),
    "duo_int4_quantize_batched": (
        ctypes.c_int,
        [c_void_p, c_int32, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_int32, c_void_p],
    ),
    "duo_int4_dequantize_batched_f16": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
         c_void_p],
    ),
    "duo_int4_stream_compress_batched": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
         POINTER(c_int32), c_void_p],
    ),
    "duo_attn_decode_int4_batched_f16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, POINTER(Int4Pool), POINTER(Int4Pool),
         c_float, c_int32, c_int32, c_void_p, c_int64, c_void_p],
    ),
    "duo_attn_decode_int4_f16": (
        ctypes.c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_int32, POINTER(Int4Pool), POINTER(Int4Pool), c_float, c_int32,
         c_int32, c_void_p, c_int64, c_void_p],
    ),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
# DUO_DEBUG_FLAGS=<bits>: debug flags OR-ed into every set_debug_flags call (same-box A/B of kernel variants through
# unmodified harnesses; see duo_set_debug_flags in include/duo_attn_hip.h)
_ENV_DEBUG_FLAGS = int(os.environ.get("DUO_DEBUG_FLAGS", "0") or "0", 0)


def load_library(path: Optional[str] = None):
    """Load (once) and type the C-ABI library.  Raises DuoHipError if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DUO_ATTN_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise DuoHipError(
            f"DuoAttention HIP library not found at {p}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C duo-attention_amd/csrc`. "
            "There is no CPU fallback for the attention hot path."
        )
    lib = ctypes.CDLL(p)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => symbol missing from the build
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.duo_abi_version() != ABI_VERSION:
        raise DuoHipError(f"ABI mismatch: library {lib.duo_abi_version()} vs binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    if _ENV_DEBUG_FLAGS:
        lib.duo_set_debug_flags(_ENV_DEBUG_FLAGS)
    return lib


def _check(code: int, what: str):
    if code != 0:
        msg = load_library().duo_error_string(code).decode()
        raise DuoHipError(f"{what} failed: [{code}] {msg}")


_cuda_get_device = torch._C._cuda_getDevice                        # (torch.cuda.current_device() minus its lazy-init checks:
_cuda_raw_stream = torch._C._cuda_getCurrentRawStream              #  these two are called tens of times per decode step)


def _require_gpu_bf16(t: torch.Tensor, name: str, dtype=torch.bfloat16):
    if not t.is_cuda:
        raise DuoHipError(
            f"{name} is on {t.device}; the DuoAttention hot path only runs on an MI355X (HIP) device — "
            "there is no CPU fallback."
        )
    if t.dtype != dtype:
        raise DuoHipError(f"{name} must be {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise DuoHipError(f"{name}: last (head_dim) dimension must be contiguous")
    if t.get_device() != _cuda_get_device():
        raise DuoHipError(
            f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: launches go to "
            "the current device's stream — wrap the call in `with torch.cuda.device(tensor.device):` (to_device / "
            "shard_model_for_pp / shard_model_for_tp select the rank's device themselves)")


def _stream_ptr(device=None) -> int:
    """the stream the launch goes to = the current stream of the CURRENT device.  Every wrapper below launches on the
    device its tensors live on only if that is the current device (the workspaces are keyed the same way), which
    ``_require_gpu_bf16`` enforces — a process that drives several GPUs switches with ``torch.cuda.device(...)``."""
    if device is None:
        return _cuda_raw_stream(_cuda_get_device())
    return torch.cuda.current_stream(device).cuda_stream


def make_seg(k: Optional[torch.Tensor], v: Optional[torch.Tensor], dtype=torch.bfloat16) -> KVSeg:
    """k, v: views [T, h, D] (any token/head stride, D contiguous), or [B, T, h, D] for the batched entry points,
    or None for an empty segment."""
    s = KVSeg()
    batched = k is not None and k.dim() == 4
    if k is None or k.shape[-3] == 0 or k.shape[-2] == 0:
        s.k = None
        s.v = None
        s.token_stride = 0
        s.head_stride = 0
        s.len = 0
        s.batch_stride = 0
        return s
    _require_gpu_bf16(k, "k segment", dtype)
    _require_gpu_bf16(v, "v segment", dtype)
    assert k.dim() in (3, 4) and v.shape == k.shape and k.stride() == v.stride() and k.shape[-1] == HEAD_DIM
    s.k = k.data_ptr()
    s.v = v.data_ptr()
    s.token_stride = k.stride(-3)
    s.head_stride = k.stride(-2)
    s.len = k.shape[-3]
    s.batch_stride = k.stride(0) if batched else 0
    return s


def make_class(n_kv_heads: int, q_head_offset: int, segA: KVSeg, segB: KVSeg) -> HeadClass:
    c = HeadClass()
    c.n_kv_heads = int(n_kv_heads)
    c.q_head_offset = int(q_head_offset)
    c.segA = segA
    c.segB = segB
    return c


# ----------------------------------------------------------------------------- ops
def rope_inplace(q: torch.Tensor, k: torch.Tensor, pos0: int, rope_scale: float, rope_theta: float):
    """q: [S, Hq, D], k: [S, Hkv, D] views, rotated in place (flashinfer apply_rope_inplace semantics); bf16, or
    fp16 (the INT4-KV path's model)."""
    lib = load_library()
    f16 = q.dtype == torch.float16
    _require_gpu_bf16(q, "q", q.dtype if f16 else torch.bfloat16)
    _require_gpu_bf16(k, "k", q.dtype if f16 else torch.bfloat16)
    assert q.dim() == 3 and k.dim() == 3 and q.shape[0] == k.shape[0]
    fn = lib.duo_rope_inplace_f16 if f16 else lib.duo_rope_inplace_bf16
    _check(
        fn(q.data_ptr(), q.stride(0), q.stride(1), q.shape[1], k.data_ptr(), k.stride(0), k.stride(1),
           k.shape[1], q.shape[0], int(pos0), float(rope_scale), float(rope_theta), q.shape[2], _stream_ptr()),
        "duo_rope_inplace_f16" if f16 else "duo_rope_inplace_bf16",
    )


def rope_inplace_batched(q: torch.Tensor, k: torch.Tensor, pos0, rope_scale: float, rope_theta: float):
    """q [B, S, Hq, D], k [B, S, Hkv, D] rotated in place; ``pos0``: one first position per batch row (list) or one int
    for all rows.  Equal positions = ONE launch for the whole batch."""
    lib = load_library()
    f16 = q.dtype == torch.float16
    _require_gpu_bf16(q, "q", q.dtype if f16 else torch.bfloat16)
    _require_gpu_bf16(k, "k", q.dtype if f16 else torch.bfloat16)
    assert q.dim() == 4 and k.dim() == 4 and q.shape[:2] == k.shape[:2]
    B = q.shape[0]
    rows = [int(pos0)] * B if not isinstance(pos0, (list, tuple)) else [int(p) for p in pos0]
    assert len(rows) == B
    arr = (c_int64 * B)(*rows)
    fn = lib.duo_rope_inplace_batched_f16 if f16 else lib.duo_rope_inplace_batched_bf16
    _check(fn(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), q.shape[2], k.data_ptr(), k.stride(0), k.stride(1),
              k.stride(2), k.shape[2], B, q.shape[1], arr, float(rope_scale), float(rope_theta), q.shape[3],
              _stream_ptr()), "duo_rope_inplace_batched")


def kv_append_batched(k_src, v_src, k_pool, v_pool, dst_row0: int):
    """src [B, S, h, D], pools [B, T, h, D] views: pool[:, dst_row0:dst_row0+S] = src in one launch."""
    lib = load_library()
    if k_src.shape[0] == 0 or k_src.shape[1] == 0 or k_src.shape[2] == 0:
        return
    for t, n in ((k_src, "k_src"), (v_src, "v_src"), (k_pool, "k_pool"), (v_pool, "v_pool")):
        _require_gpu_bf16(t, n)
    assert k_src.dim() == 4 and k_src.stride() == v_src.stride() and k_pool.stride() == v_pool.stride()
    assert k_pool.shape[0] == k_src.shape[0] and dst_row0 + k_src.shape[1] <= k_pool.shape[1]
    _check(
        lib.duo_kv_append_batched_bf16(
            k_src.data_ptr(), v_src.data_ptr(), k_src.stride(0), k_src.stride(1), k_src.stride(2), k_pool.data_ptr(),
            v_pool.data_ptr(), k_pool.stride(0), k_pool.stride(1), k_pool.stride(2), k_src.shape[0], k_src.shape[2],
            k_src.shape[1], int(dst_row0), k_src.shape[3], _stream_ptr(),
        ),
        "duo_kv_append_batched_bf16",
    )


def stream_compress_batched(k_pool, v_pool, k_new, v_new, cur_len: int, sink: int, recent: int) -> int:
    """pools [B, W, h, D], new rows [B, S, h, D]: the sink+recent update of every batch row in one launch."""
    lib = load_library()
    new_len = c_int32(0)
    n_heads = k_pool.shape[2]
    live = n_heads > 0 and k_new.shape[1] > 0 and k_pool.shape[0] > 0
    if live:
        for t, n in ((k_pool, "k_pool"), (v_pool, "v_pool"), (k_new, "k_new"), (v_new, "v_new")):
            _require_gpu_bf16(t, n)
        assert k_new.stride() == v_new.stride() and k_pool.stride() == v_pool.stride()
        assert k_pool.shape[1] >= sink + recent and k_pool.shape[0] == k_new.shape[0]
    _check(
        lib.duo_stream_compress_batched_bf16(
            k_pool.data_ptr() if live else None, v_pool.data_ptr() if live else None, k_pool.stride(0), k_pool.stride(1),
            k_pool.stride(2), k_new.data_ptr() if live else None, v_new.data_ptr() if live else None, k_new.stride(0),
            k_new.stride(1), k_new.stride(2), k_pool.shape[0], n_heads, int(cur_len), k_new.shape[1], int(sink),
            int(recent), HEAD_DIM, byref(new_len), _stream_ptr(),
        ),
        "duo_stream_compress_batched_bf16",
    )
    return int(new_len.value)


def kv_append(k_src, v_src, k_pool, v_pool, dst_row0: int):
    """src: [S, h, D] views; pools: [T, h, D] views.  pool[dst_row0:dst_row0+S] = src."""
    lib = load_library()
    if k_src.shape[0] == 0 or k_src.shape[1] == 0:
        return
    for t, n in ((k_src, "k_src"), (v_src, "v_src"), (k_pool, "k_pool"), (v_pool, "v_pool")):
        _require_gpu_bf16(t, n)
    assert k_src.stride() == v_src.stride() and k_pool.stride() == v_pool.stride()
    assert dst_row0 + k_src.shape[0] <= k_pool.shape[0]
    _check(
        lib.duo_kv_append_bf16(
            k_src.data_ptr(), v_src.data_ptr(), k_src.stride(0), k_src.stride(1), k_pool.data_ptr(),
            v_pool.data_ptr(), k_pool.stride(0), k_pool.stride(1), k_src.shape[1], k_src.shape[0],
            int(dst_row0), k_src.shape[2], _stream_ptr(),
        ),
        "duo_kv_append_bf16",
    )


def stream_compress(k_pool, v_pool, k_new, v_new, cur_len: int, sink: int, recent: int) -> int:
    """In-place sink+recent update of a streaming pool [W, h, D] with new rows [S, h, D]; returns new length."""
    lib = load_library()
    new_len = c_int32(0)
    n_heads = k_pool.shape[1]
    if n_heads > 0 and k_new.shape[0] > 0:
        for t, n in ((k_pool, "k_pool"), (v_pool, "v_pool"), (k_new, "k_new"), (v_new, "v_new")):
            _require_gpu_bf16(t, n)
        assert k_new.stride() == v_new.stride() and k_pool.stride() == v_pool.stride()
        assert k_pool.shape[0] >= sink + recent
    _check(
        lib.duo_stream_compress_bf16(
            k_pool.data_ptr() if n_heads else None, v_pool.data_ptr() if n_heads else None,
            k_pool.stride(0), k_pool.stride(1), k_new.data_ptr() if n_heads else None,
            v_new.data_ptr() if n_heads else None, k_new.stride(0), k_new.stride(1), n_heads, int(cur_len),
            k_new.shape[0], int(sink), int(recent), HEAD_DIM, byref(new_len), _stream_ptr(),
        ),
        "duo_stream_compress_bf16",
    )
    return int(new_len.value)


_DECODE_MAX_SPLITS = 512
DECODE_TICKET_BYTES = 4096      # DUO_DECODE_TICKET_BYTES
_workspaces = {}
_tickets = {}
_graph_tickets = []      # weak references to the tickets owned by captured decode steps (scratch_scope)


def _stream_key(device: torch.device):
    idx = device.index if device.index is not None else _cuda_get_device()
    return (idx, _cuda_raw_stream(idx))


def decode_workspace(device: torch.device, n_q_heads: int) -> torch.Tensor:
    """fp32 scratch for the split-KV partials: one buffer per (device, stream, q-head count), allocated once and
    never replaced or freed — concurrent streams do not share partials, and a captured graph's launches keep
    pointing at live memory whatever other models decode later."""
    scope = _active_scope()
    cache = _workspaces if scope is None else scope
    key = _stream_key(device) + (int(n_q_heads),)
    ws = cache.get(key)
    if ws is None:
        need = load_library().duo_attn_decode_workspace_bytes(int(n_q_heads), _DECODE_MAX_SPLITS)
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        cache[key] = ws
    return ws


_scope_tls = threading.local()      # the active scratch_scope of THIS thread (None outside one)


def _active_scope():
    return getattr(_scope_tls, "owner", None)


class scratch_scope:
    """While active ON THIS THREAD, the decode scratch (split-KV partials, arrival tickets) comes from ``owner`` — a dict the
    caller keeps — instead of the per-(device, stream) caches.  A captured decode step uses it (duo_attn/graph.py): every graph
    captures on torch's one shared capture stream, so graphs of two caches would otherwise bake the SAME partials buffer into
    their launches.  Thread-local: an eager decode on another thread (another device, another stream) keeps its own scratch
    while this thread captures.  What the scope does NOT cover: torch's GEMM workspace (per capture stream) — two captured
    whole-model steps must not be replayed concurrently on two streams."""

    def __init__(self, owner: dict):
        self.owner = owner

    def __enter__(self):
        self.prev, _scope_tls.owner = _active_scope(), self.owner
        return self.owner

    def __exit__(self, *exc):
        _scope_tls.owner = self.prev
        return False


def prepare_graph_scratch(owner: dict, kv_cache) -> None:
    """Called by ``DecodeStepGraph.__init__`` OUTSIDE the capture: the arrival tickets of the (opt-in) single-launch decode step
    exist and are zero before the first launch is recorded, so no memset is part of the graph — a replay can never clear the
    sticky give-up flag, and ``check_decode_tickets`` never reads uninitialised memory of a captured-but-never-replayed step.
    (The split-KV partials need no initial value: they are allocated on first use inside the scope.)"""
    if "tickets" not in owner:
        t = owner["tickets"] = torch.zeros(DECODE_TICKET_BYTES // 4, dtype=torch.int32, device=kv_cache.device)
        _graph_tickets.append(weakref.ref(t))


def release_workspaces(device=None) -> None:
    """Free the cached scratch buffers (split-KV partials ~8.5 MB, prefill key-range partials, arrival tickets) of one
    device, or of all.  They are kept per (device, stream) so that concurrent streams never share partials and a
    captured graph keeps pointing at live memory: call this only when no graph captured on those streams will be
    replayed again (a process that rotates through many streams would otherwise accumulate one set per stream)."""
    idx = None if device is None else (torch.device(device).index if torch.device(device).index is not None
                                       else torch.cuda.current_device())
    for cache in (_workspaces, _tickets, _prefill_ws):
        for key in [k for k in cache if idx is None or k[0] == idx]:
            del cache[key]


def check_decode_tickets(device=None) -> None:
    """Single-launch decode step (opt-in): every launch must leave the arrival tickets zeroed and the give-up flag clear.
    One device read-back — call it at sequence boundaries, not per step (``DuoAttentionStaticKVCache.clear`` does when the
    single-launch step was used).  A set flag means a merger stopped waiting and merged incomplete partials: the tickets
    are re-armed and DuoHipError is raised."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    bad = []
    _graph_tickets[:] = [r for r in _graph_tickets if r() is not None]
    graph_owned = [((idx, "graph"), t) for t in (r() for r in _graph_tickets) if t is not None and t.device.index == idx]
    for key, t in list(_tickets.items()) + graph_owned:
        if key[0] == idx and int(t.abs().sum()) != 0:
            bad.append((key, int(t[-1])))
            t.zero_()
    if bad:
        raise DuoHipError(f"single-launch decode step left its arrival tickets non-zero (stream, give-up flag): {bad}; "
                          "results of that sequence are not trustworthy — use the default two-launch step")


_one_launch_used = False


def decode_tickets(device: torch.device) -> torch.Tensor:
    """Arrival tickets of the single-launch decode step (duo_decode_step_bf16): zero-filled once per
    (device, stream); every launch leaves them zeroed."""
    scope = _active_scope()
    if scope is not None:
        t = scope.get("tickets")
        if t is None:       # (a scope that was not prepared by prepare_graph_scratch: a direct user outside any capture)
            if torch.cuda.is_current_stream_capturing():
                raise DuoHipError("decode tickets requested inside a stream capture without prepare_graph_scratch()")
            t = scope["tickets"] = torch.zeros(DECODE_TICKET_BYTES // 4, dtype=torch.int32, device=device)
            _graph_tickets.append(weakref.ref(t))
        return t
    key = _stream_key(device)
    t = _tickets.get(key)
    if t is None:
        t = torch.zeros(DECODE_TICKET_BYTES // 4, dtype=torch.int32, device=device)
        _tickets[key] = t
    return t


def attn_decode(q: torch.Tensor, out: torch.Tensor, group: int, full: Optional[HeadClass],
                stream: Optional[HeadClass], scale: float):
    """q, out: [Hq, D] views (one token).  Split-KV decode over both head classes."""
    lib = load_library()
    _require_gpu_bf16(q, "q")
    _require_gpu_bf16(out, "out")
    assert q.dim() == 2 and out.shape == q.shape
    ws = decode_workspace(q.device, q.shape[0])
    _check(
        lib.duo_attn_decode_bf16(
            q.data_ptr(), q.stride(0), out.data_ptr(), out.stride(0), int(group),
            byref(full) if full is not None else None, byref(stream) if stream is not None else None,
            float(scale), q.shape[1], ws.data_ptr(), ws.numel() * 4, _stream_ptr(),
        ),
        "duo_attn_decode_bf16",
    )


def attn_decode_tuple(q, out, group: int, n_full: int, arena, full_len: int, str_src, k, v, scale: float):
    """The two flash_attn_func calls of the tuple-cache forward at q_len == 1 (reference llama.py:225-262) as ONE split-KV
    decode launch pair, straight over the tuple format: retrieval heads over rows [0, full_len) of ``arena`` ([2, nf, cap, D]:
    K then V) ++ the new row, streaming heads over ``str_src`` ([2, ns, n, D]) ++ the new row; ``k`` / ``v`` [Hkv, D] are the
    new (rotated) rows, ``q`` / ``out`` [Hq, D].  Same kernel and arithmetic as ``attn_decode`` — this form only spares the
    host the dozen tensor views per layer and token that describing the segments as tensors costs."""
    lib = load_library()
    for t, n in ((q, "q"), (out, "out"), (k, "k"), (v, "v")):
        _require_gpu_bf16(t, n)
    Hq, D = q.shape
    Hkv = k.shape[0]
    ns = Hkv - n_full
    khs = k.stride(0)
    assert v.stride(0) == khs and out.shape == q.shape
    fc = sc = None
    if n_full > 0:
        _require_gpu_bf16(arena, "arena")
        fc = HeadClass()
        fc.n_kv_heads, fc.q_head_offset = n_full, 0
        a, b = fc.segA, fc.segB
        if full_len > 0:
            a.k, a.v = arena.data_ptr(), arena.data_ptr() + 2 * arena.stride(0)
            a.token_stride, a.head_stride, a.len = arena.stride(2), arena.stride(1), int(full_len)
        b.k, b.v, b.token_stride, b.head_stride, b.len = k.data_ptr(), v.data_ptr(), 0, khs, 1
    if ns > 0:
        sc = HeadClass()
        sc.n_kv_heads, sc.q_head_offset = ns, n_full * int(group)
        a, b = sc.segA, sc.segB
        n = str_src.shape[2]
        if n > 0:
            _require_gpu_bf16(str_src, "streaming cache")
            a.k, a.v = str_src.data_ptr(), str_src.data_ptr() + 2 * str_src.stride(0)
            a.token_stride, a.head_stride, a.len = str_src.stride(2), str_src.stride(1), n
        off = 2 * n_full * khs
        b.k, b.v, b.token_stride, b.head_stride, b.len = k.data_ptr() + off, v.data_ptr() + off, 0, khs, 1
    ws = decode_workspace(q.device, Hq)
    _check(lib.duo_attn_decode_bf16(q.data_ptr(), q.stride(0), out.data_ptr(), out.stride(0), int(group),
                                    byref(fc) if fc is not None else None, byref(sc) if sc is not None else None,
                                    float(scale), D, ws.data_ptr(), ws.numel() * 4, _stream_ptr()), "duo_attn_decode_bf16")


def _decode_layer_args(q, k, v, out, n_full, full_k, full_v, full_len, str_k, str_v, str_len, sink, recent, pos,
                       rope_scale, rope_theta, scale) -> DecodeLayerArgs:
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
        _require_gpu_bf16(t, n)
    assert k.stride(0) == v.stride(0)
    a = DecodeLayerArgs()
    a.q, a.q_head_stride, a.n_q_heads, a.n_kv_heads = q.data_ptr(), q.stride(0), q.shape[0], k.shape[0]
    a.k, a.v, a.kv_head_stride = k.data_ptr(), v.data_ptr(), k.stride(0)
    a.out, a.out_head_stride = out.data_ptr(), out.stride(0)
    a.n_full, a.head_dim = int(n_full), q.shape[1]
    ns = k.shape[0] - n_full
    if n_full > 0:
        _require_gpu_bf16(full_k, "full_k")
        assert full_k.stride() == full_v.stride()
        a.full_k, a.full_v = full_k.data_ptr(), full_v.data_ptr()
        a.full_token_stride, a.full_head_stride = full_k.stride(0), full_k.stride(1)
        a.full_capacity = full_k.shape[0]
    a.full_len = int(full_len)
    if ns > 0:
        _require_gpu_bf16(str_k, "str_k")
        assert str_k.stride() == str_v.stride() and str_k.shape[0] >= sink + recent
        a.str_k, a.str_v = str_k.data_ptr(), str_v.data_ptr()
        a.str_token_stride, a.str_head_stride = str_k.stride(0), str_k.stride(1)
    a.str_len, a.sink, a.recent = int(str_len), int(sink), int(recent)
    a.pos, a.rope_scale, a.rope_theta, a.scale = int(pos), float(rope_scale), float(rope_theta), float(scale)
    return a


# The single-launch step (duo_decode_step_bf16) is bit-identical to the two-launch form.  Which one is faster depends on the
# context (round 6, GPU-side time of the 32-layer step as a captured graph, profiles/r6_decode_short.md): at <= 16K cached
# rows a layer's scan is 10 us and the second launch with the kernel boundary in front of it is a fifth of the step — ONE
# launch is 10-18 % faster (0.41 vs 0.49 ms/token at 4K, 0.50 vs 0.55 at 16K) and saves a launch on the host-bound eager loop
# as well; at 32K they are equal; from 64K on publishing partials to other workgroups of the same launch costs more than the
# boundary it removes (round 2: +1.3 us per layer at 128K, profiles/r2_decode_one_launch.md).  So the form follows the length:
# one launch up to ONE_LAUNCH_MAX_ROWS cached rows — a bucket boundary of the split-KV planner, so a captured step and the
# eager step of the same length always agree on the form — two above.  DUO_DECODE_ONE_LAUNCH=1 / 0 forces one / two.
ONE_LAUNCH_MAX_ROWS = 16384
_ONE_LAUNCH_ENV = os.environ.get("DUO_DECODE_ONE_LAUNCH")


def _two_launch(full_len: int, str_len: int, two_launch) -> bool:
    if two_launch is not None:
        return bool(two_launch)
    if _ONE_LAUNCH_ENV in ("0", "1"):
        return _ONE_LAUNCH_ENV == "0"
    return max(int(full_len), int(str_len)) + 1 > ONE_LAUNCH_MAX_ROWS


def decode_layer(q, k, v, out, n_full, full_k, full_v, full_len, str_k, str_v, str_len, sink, recent, pos,
                 rope_scale, rope_theta, scale, two_launch: Optional[bool] = None) -> int:
    """Fused decode step of one layer (one batch row): scan + epilogue launch pair, or everything in ONE launch
    (``two_launch``: None = by context length, see the note above).  q/out [Hq, D]; k/v [Hkv, D] new rows; full_k/full_v [T, nf, D]
    and str_k/str_v [W, ns, D] pool views.  Returns the new streaming length."""
    lib = load_library()
    a = _decode_layer_args(q, k, v, out, n_full, full_k, full_v, full_len, str_k, str_v, str_len, sink, recent, pos,
                           rope_scale, rope_theta, scale)
    ws = decode_workspace(q.device, q.shape[0])
    new_len = c_int32(0)
    if _two_launch(full_len, str_len, two_launch):
        _check(lib.duo_decode_layer_bf16(byref(a), byref(new_len), ws.data_ptr(), ws.numel() * 4, _stream_ptr()),
               "duo_decode_layer_bf16")
    else:
        global _one_launch_used
        _one_launch_used = True
        _check(lib.duo_decode_step_bf16(byref(a), byref(new_len), None, ws.data_ptr(), ws.numel() * 4,
                                        decode_tickets(q.device).data_ptr(), _stream_ptr()), "duo_decode_step_bf16")
    return int(new_len.value)


def decode_layer_dev(q, k, v, out, n_full, full_k, full_v, plan_full_len, str_k, str_v, plan_str_len, sink, recent,
                     plan_pos, rope_scale, rope_theta, scale, dev_state: torch.Tensor,
                     two_launch: Optional[bool] = None) -> None:
    """The same step with lengths / position read from ``dev_state`` (int32 [4] on the GPU:
    full_len, str_len, pos, pad) — graph-capturable; the ``plan_*`` values only size the grid."""
    lib = load_library()
    assert dev_state.is_cuda and dev_state.dtype == torch.int32 and dev_state.numel() >= 4 and dev_state.is_contiguous()
    a = _decode_layer_args(q, k, v, out, n_full, full_k, full_v, plan_full_len, str_k, str_v, plan_str_len, sink,
                           recent, plan_pos, rope_scale, rope_theta, scale)
    ws = decode_workspace(q.device, q.shape[0])
    if _two_launch(plan_full_len, plan_str_len, two_launch):
        _check(lib.duo_decode_layer_dev_bf16(byref(a), dev_state.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                             _stream_ptr()), "duo_decode_layer_dev_bf16")
    else:
        global _one_launch_used
        _one_launch_used = True
        _check(lib.duo_decode_step_bf16(byref(a), None, dev_state.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                        decode_tickets(q.device).data_ptr(), _stream_ptr()), "duo_decode_step_bf16")


def decode_state_add(dev_states: torch.Tensor, d_full: int, d_str: int, d_pos: int, str_cap: int) -> None:
    """dev_states: int32 [n_layers, 4] on the GPU; one launch advances / rewinds every layer's counters."""
    lib = load_library()
    assert dev_states.is_cuda and dev_states.dtype == torch.int32 and dev_states.is_contiguous() and dev_states.shape[-1] == 4
    _check(lib.duo_decode_state_add(dev_states.data_ptr(), dev_states.numel() // 4, int(d_full), int(d_str), int(d_pos),
                                    int(str_cap), _stream_ptr()), "duo_decode_state_add")


_prefill_ws = {}


def prefill_workspace(device) -> torch.Tensor:
    """fp32 workspace for the key-range splits of the prefill kernel: one per (device, stream), allocated once."""
    key = _stream_key(torch.device(device))
    ws = _prefill_ws.get(key)
    if ws is None:
        ws = torch.empty(load_library().duo_attn_prefill_workspace_bytes() // 4, dtype=torch.float32, device=device)
        _prefill_ws[key] = ws
    return ws


def attn_prefill(q: torch.Tensor, out: torch.Tensor, group: int, full: Optional[HeadClass],
                 stream: Optional[HeadClass], scale: float):
    """q, out: [S, Hq, D] views, bf16 or fp16 (the segments of ``full`` / ``stream`` must have the same
    element type).  MFMA flash attention over both head classes."""
    lib = load_library()
    if q.dtype not in (torch.bfloat16, torch.float16):
        raise DuoHipError(f"q must be bfloat16 or float16, got {q.dtype}")
    _require_gpu_bf16(q, "q", q.dtype)
    _require_gpu_bf16(out, "out", q.dtype)
    assert q.dim() == 3 and out.shape == q.shape
    fn = lib.duo_attn_prefill_ws_f16 if q.dtype == torch.float16 else lib.duo_attn_prefill_ws_bf16
    ws = prefill_workspace(q.device)
    _check(
        fn(
            q.data_ptr(), q.stride(0), q.stride(1), out.data_ptr(), out.stride(0), out.stride(1), q.shape[0],
            int(group), byref(full) if full is not None else None,
            byref(stream) if stream is not None else None, float(scale), q.shape[2],
            ws.data_ptr(), ws.numel() * 4, _stream_ptr(),
        ),
        "duo_attn_prefill_ws_f16" if q.dtype == torch.float16 else "duo_attn_prefill_ws_bf16",
    )


def attention_batched(q: torch.Tensor, out: torch.Tensor, group: int, full: Optional[HeadClass],
                      stream: Optional[HeadClass], scale: float):
    """q, out: [B, S, Hq, D] views; the classes' segments were built from [B, T, h, D] views (``make_seg``).  S > 1 (and
    fp16 at any S): MFMA prefill, bf16 S == 1: split-KV decode — one launch (pair) for all batch rows."""
    lib = load_library()
    if q.dtype not in (torch.bfloat16, torch.float16):
        raise DuoHipError(f"q must be bfloat16 or float16, got {q.dtype}")
    _require_gpu_bf16(q, "q", q.dtype)
    _require_gpu_bf16(out, "out", q.dtype)
    assert q.dim() == 4 and out.shape == q.shape
    B, S = q.shape[0], q.shape[1]
    fc, sc = (byref(full) if full is not None else None), (byref(stream) if stream is not None else None)
    if S == 1 and q.dtype == torch.bfloat16:       # (fp16 single rows: a one-row query block of the MFMA kernel, below)
        ws = decode_workspace(q.device, q.shape[2])      # every batch row gets its own share of the partial area
        _check(lib.duo_attn_decode_batched_bf16(q.data_ptr(), q.stride(0), q.stride(2), out.data_ptr(), out.stride(0),
                                                out.stride(2), B, int(group), fc, sc, float(scale), q.shape[3],
                                                ws.data_ptr(), ws.numel() * 4, _stream_ptr()),
               "duo_attn_decode_batched_bf16")
        return
    fn = lib.duo_attn_prefill_batched_f16 if q.dtype == torch.float16 else lib.duo_attn_prefill_batched_bf16
    ws = prefill_workspace(q.device)
    _check(fn(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), out.data_ptr(), out.stride(0), out.stride(1),
              out.stride(2), B, S, int(group), fc, sc, float(scale), q.shape[3], ws.data_ptr(), ws.numel() * 4,
              _stream_ptr()), "duo_attn_prefill_batched")


def decode_layer_batched(q, k, v, out, n_full, full_k, full_v, full_len, str_k, str_v, str_len, sink, recent, pos,
                         rope_scale, rope_theta, scale) -> int:
    """The fused decode step of one layer for ALL batch rows in one launch pair: q/out [B, Hq, D]; k/v [B, Hkv, D] new
    rows; full_k/full_v [B, T, nf, D], str_k/str_v [B, W, ns, D] pool views; ``pos``: one int or one per row."""
    lib = load_library()
    B = q.shape[0]
    a = _decode_layer_args(q[0], k[0], v[0], out[0], n_full, full_k[0] if n_full > 0 else None,
                           full_v[0] if n_full > 0 else None, full_len, str_k[0] if k.shape[1] - n_full > 0 else None,
                           str_v[0] if k.shape[1] - n_full > 0 else None, str_len, sink, recent,
                           pos[0] if isinstance(pos, (list, tuple)) else pos, rope_scale, rope_theta, scale)
    rows = [int(pos)] * B if not isinstance(pos, (list, tuple)) else [int(p) for p in pos]
    arr = (c_int64 * B)(*rows)
    bt = DecodeBatch()
    bt.n_batch = B
    bt.q_batch_stride, bt.kv_batch_stride, bt.out_batch_stride = q.stride(0), k.stride(0), out.stride(0)
    assert v.stride(0) == k.stride(0)
    bt.full_batch_stride = full_k.stride(0) if n_full > 0 else 0
    bt.str_batch_stride = str_k.stride(0) if k.shape[1] - n_full > 0 else 0
    bt.pos = arr
    ws = decode_workspace(q.device, q.shape[1])
    new_len = c_int32(0)
    _check(lib.duo_decode_layer_batched_bf16(byref(a), byref(bt), byref(new_len), ws.data_ptr(), ws.numel() * 4,
                                             _stream_ptr()), "duo_decode_layer_batched_bf16")
    return int(new_len.value)


def decode_layer_batched_dev(q, k, v, out, n_full, full_k, full_v, plan_full_len, str_k, str_v, plan_str_len, sink, recent,
                             pos, rope_scale, rope_theta, scale, dev_state: torch.Tensor) -> None:
    """``decode_layer_batched`` with the lengths / position read from ``dev_state`` (int32 [4] on the GPU) — graph-capturable
    for B > 1.  ``pos``: one int, or one per row: row b runs at ``dev_state.pos + (pos[b] - plan_full_len)`` (the device
    counter starts as the cache length; a row's offset from it is fixed for the life of a sequence); the ``plan_*`` values
    only size the grid."""
    lib = load_library()
    assert dev_state.is_cuda and dev_state.dtype == torch.int32 and dev_state.numel() >= 4 and dev_state.is_contiguous()
    B = q.shape[0]
    ns = k.shape[1] - n_full
    # args.pos = the host's view of the device-side position counter (= the cache length, sync_device_state)
    a = _decode_layer_args(q[0], k[0], v[0], out[0], n_full, full_k[0] if n_full > 0 else None,
                           full_v[0] if n_full > 0 else None, plan_full_len, str_k[0] if ns > 0 else None,
                           str_v[0] if ns > 0 else None, plan_str_len, sink, recent,
                           plan_full_len, rope_scale, rope_theta, scale)
    rows = [int(pos)] * B if not isinstance(pos, (list, tuple)) else [int(p) for p in pos]
    arr = (c_int64 * B)(*rows)
    bt = DecodeBatch()
    bt.n_batch = B
    bt.q_batch_stride, bt.kv_batch_stride, bt.out_batch_stride = q.stride(0), k.stride(0), out.stride(0)
    assert v.stride(0) == k.stride(0)
    bt.full_batch_stride = full_k.stride(0) if n_full > 0 else 0
    bt.str_batch_stride = str_k.stride(0) if ns > 0 else 0
    bt.pos = arr
    ws = decode_workspace(q.device, q.shape[1])
    _check(lib.duo_decode_layer_batched_dev_bf16(byref(a), byref(bt), dev_state.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                 _stream_ptr()), "duo_decode_layer_batched_dev_bf16")


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """flashinfer.norm.rmsnorm semantics on [rows, hidden] bf16."""
    lib = load_library()
    _require_gpu_bf16(x, "x")
    _require_gpu_bf16(weight, "weight")
    x2 = x.contiguous().view(-1, x.shape[-1])
    y = torch.empty_like(x2)
    _check(
        lib.duo_rmsnorm_bf16(x2.data_ptr(), weight.contiguous().data_ptr(), y.data_ptr(), x2.shape[0],
                             x2.shape[1], float(eps), _stream_ptr()),
        "duo_rmsnorm_bf16",
    )
    return y.view(x.shape)


def rmsnorm_hf(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """transformers LlamaRMSNorm / MistralRMSNorm.forward on [..., hidden] bf16 in one pass: the normalised activations are
    rounded to bf16 before the weight multiply (two roundings; ``rmsnorm`` is flashinfer's one-rounding form)."""
    lib = load_library()
    _require_gpu_bf16(x, "x")
    _require_gpu_bf16(weight, "weight")
    x2 = x.contiguous().view(-1, x.shape[-1])
    w = weight if weight.is_contiguous() else weight.contiguous()
    y = torch.empty_like(x2)
    _check(lib.duo_rmsnorm_hf_bf16(x2.data_ptr(), w.data_ptr(), y.data_ptr(), x2.shape[0], x2.shape[1], float(eps),
                                   _stream_ptr()), "duo_rmsnorm_hf_bf16")
    return y.view(x.shape)


def rope_hf_inplace(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> None:
    """transformers ``apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=2)`` IN PLACE on q [S, Hq, D] and k [S, Hkv, D]
    (views, D contiguous); cos / sin [S, D] bf16 — the rows ``model.rotary_emb`` computed for these positions.  torch's bf16
    arithmetic (each product and the sum rounded to bf16): bit-equal to the six elementwise kernels it replaces."""
    lib = load_library()
    for t, n in ((q, "q"), (k, "k"), (cos, "cos"), (sin, "sin")):
        _require_gpu_bf16(t, n)
    S, D = q.shape[0], q.shape[2]
    assert k.shape[0] == S and cos.shape == (S, D) and sin.shape == (S, D) and cos.stride(0) == sin.stride(0)
    _check(lib.duo_rope_hf_inplace_bf16(q.data_ptr(), q.stride(0), q.stride(1), q.shape[1], k.data_ptr(), k.stride(0),
                                        k.stride(1), k.shape[1], S, cos.data_ptr(), sin.data_ptr(), cos.stride(0), D,
                                        _stream_ptr()), "duo_rope_hf_inplace_bf16")


def silu_mul(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """``silu(gate) * up`` (HF LlamaMLP's activation product) in one pass; [..., n] bf16, unit inner stride."""
    lib = load_library()
    _require_gpu_bf16(gate, "gate")
    _require_gpu_bf16(up, "up")
    if gate.shape != up.shape or gate.shape[-1] % 8:
        raise DuoHipError(f"silu_mul: shapes {tuple(gate.shape)} / {tuple(up.shape)}")
    g2, u2 = gate.reshape(-1, gate.shape[-1]), up.reshape(-1, up.shape[-1])
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    if u2.stride(1) != 1:
        u2 = u2.contiguous()
    y = torch.empty(g2.shape, dtype=torch.bfloat16, device=gate.device)
    _check(lib.duo_silu_mul_bf16(g2.data_ptr(), g2.stride(0), u2.data_ptr(), u2.stride(0), y.data_ptr(), y.stride(0),
                                 g2.shape[0], g2.shape[1], _stream_ptr()), "duo_silu_mul_bf16")
    return y.view(gate.shape)


def token_linear_fits(n_rows: int, n_in: int) -> bool:
    """whether duo_token_linear_bf16 takes ``n_rows`` token rows of ``n_in`` features (rows in LDS, 156 KiB)"""
    kpad = -(-n_in // TOKEN_LINEAR_PAD) * TOKEN_LINEAR_PAD
    return 1 <= n_rows <= TOKEN_LINEAR_MAX_ROWS and n_in >= 8 and n_in % 8 == 0 and n_rows * kpad * 2 <= 156 * 1024


def token_linear(x: torch.Tensor, blocks, norm=None, x2: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None, norm_hf: bool = False) -> torch.Tensor:
    """Token-row linear layers of the decode step in ONE launch (duo_token_linear_bf16):
    ``y = [W0; W1; W2] @ xn + bias (+ residual)`` with ``xn = x``, ``rmsnorm(x; *norm)`` or ``silu(x) * x2``.
    x [rows, n_in] bf16 (rows <= 4, unit inner stride); blocks: up to three ``(weight [n, n_in], bias or None)``;
    norm: ``(weight [n_in], eps)`` — the flashinfer form (one rounding; the static path's norm), or with ``norm_hf`` the
    HuggingFace ``*RMSNorm.forward`` form (normalised x rounded to bf16 before the weight multiply; the tuple path's);
    returns [rows, sum n] bf16."""
    lib = load_library()
    _require_gpu_bf16(x, "x")
    rows, n_in = x.shape
    if x.stride(1) != 1 or not token_linear_fits(rows, n_in) or not 1 <= len(blocks) <= 3:
        raise DuoHipError(f"token_linear: unsupported input {tuple(x.shape)} strides {x.stride()} with {len(blocks)} blocks")
    a = TokenLinearArgs()
    a.x, a.x_row_stride, a.n_rows, a.n_in = x.data_ptr(), x.stride(0), rows, n_in
    if x2 is not None:
        _require_gpu_bf16(x2, "x2")
        if x2.shape != x.shape or x2.stride() != x.stride():
            raise DuoHipError("token_linear: x2 must have the shape and strides of x")
        a.x2 = x2.data_ptr()
    n_total = 0
    keep = []       # contiguous copies of bias / norm weight stay referenced until the launch is enqueued: a temporary freed
    #                 before `y` is allocated could be handed to `y` by the caching allocator (or a graph's private pool)
    for i, (w, b) in enumerate(blocks):
        _require_gpu_bf16(w, "weight")
        if w.dim() != 2 or w.shape[1] != n_in or w.stride(1) != 1:
            raise DuoHipError(f"token_linear: weight {tuple(w.shape)} strides {w.stride()} does not match n_in {n_in}")
        a.seg[i].w, a.seg[i].row_stride, a.seg[i].n = w.data_ptr(), w.stride(0), w.shape[0]
        if b is not None:
            _require_gpu_bf16(b, "bias")
            bc = b if b.is_contiguous() else b.contiguous()
            keep.append(bc)
            a.seg[i].bias = bc.data_ptr()
        n_total += w.shape[0]
    if norm is not None:
        _require_gpu_bf16(norm[0], "norm weight")
        nw = norm[0] if norm[0].is_contiguous() else norm[0].contiguous()
        keep.append(nw)
        a.norm_weight, a.norm_eps = nw.data_ptr(), float(norm[1])
        a.flags = LINEAR_NORM_HF if norm_hf else 0
    y = torch.empty(rows, n_total, dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        _require_gpu_bf16(residual, "residual")
        if residual.shape != y.shape or residual.stride(1) != 1:
            raise DuoHipError("token_linear: residual must be [rows, n_total] with unit inner stride")
        a.residual, a.residual_row_stride = residual.data_ptr(), residual.stride(0)
    a.y, a.y_row_stride = y.data_ptr(), y.stride(0)
    _check(lib.duo_token_linear_bf16(byref(a), _stream_ptr()), "duo_token_linear_bf16")
    del keep
    return y


def tuple_decode_prep(q, k, v, cos_row, sin_row, n_full: int, arena, full_len: int, str_src, sink: int, recent: int):
    """The tuple-cache decode step's data movement in ONE launch (duo_tuple_decode_prep_bf16; reference llama.py:177-184,
    :202-223, :273-301 at q_len == 1, one batch row): q [Hq, D] and k [Hkv, D] are rotated IN PLACE with HF's rotary in
    torch's bf16 arithmetic (cos_row / sin_row: [D] bf16); the retrieval heads' new rows land at row ``full_len`` of
    ``arena`` ([2, nf, cap, D]: K then V); the streaming cache ``str_src`` ([2, ns, n, D]) ++ new row, truncated to
    sink + recent, is written to a NEW tensor, which is returned ([2, ns, min(n + 1, sink + recent), D], head-major)."""
    lib = load_library()
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (cos_row, "cos"), (sin_row, "sin")):
        _require_gpu_bf16(t, n)
    Hq, D = q.shape
    Hkv = k.shape[0]
    ns = Hkv - n_full
    assert k.stride(0) == v.stride(0) and cos_row.numel() == D and sin_row.numel() == D
    a = TupleDecodeArgs()
    a.q, a.q_head_stride, a.n_q_heads, a.n_kv_heads = q.data_ptr(), q.stride(0), Hq, Hkv
    a.k, a.v, a.kv_head_stride = k.data_ptr(), v.data_ptr(), k.stride(0)
    a.cos_row, a.sin_row = cos_row.data_ptr(), sin_row.data_ptr()
    a.n_full, a.head_dim = int(n_full), D
    if n_full > 0:
        _require_gpu_bf16(arena, "arena")
        assert arena.dim() == 4 and arena.shape[0] == 2 and arena.shape[1] == n_full and arena.stride(3) == 1
        a.full_k, a.full_v = arena[0].data_ptr(), arena[1].data_ptr()
        a.full_token_stride, a.full_head_stride = arena.stride(2), arena.stride(1)
        a.full_capacity = arena.shape[2]
    a.full_len = int(full_len)
    n = int(str_src.shape[2])
    out_len = min(n + 1, sink + recent)
    dst = torch.empty(2, ns, out_len, D, dtype=torch.bfloat16, device=q.device)
    if ns > 0:
        if n > 0:
            _require_gpu_bf16(str_src, "streaming cache")
            assert str_src.shape[0] == 2 and str_src.shape[1] == ns
            a.str_k_src, a.str_v_src = str_src[0].data_ptr(), str_src[1].data_ptr()
            a.src_token_stride, a.src_head_stride = str_src.stride(2), str_src.stride(1)
        if out_len > 0:
            a.str_k_dst, a.str_v_dst = dst[0].data_ptr(), dst[1].data_ptr()
            a.dst_token_stride, a.dst_head_stride = dst.stride(2), dst.stride(1)
    a.str_len, a.sink, a.recent = n, int(sink), int(recent)
    new_len = c_int32(0)
    _check(lib.duo_tuple_decode_prep_bf16(byref(a), byref(new_len), _stream_ptr()), "duo_tuple_decode_prep_bf16")
    assert new_len.value == out_len
    return dst


# ----------------------------------------------------------------------------- INT4 KV pools
def _require_gpu(t: torch.Tensor, name: str, dtype):
    if not t.is_cuda:
        raise DuoHipError(f"{name} is on {t.device}; the INT4 KV path only runs on an MI355X — no CPU fallback.")
    if t.dtype != dtype:
        raise DuoHipError(f"{name} must be {dtype}, got {t.dtype}")


def _pool_row_strides(q_pool: torch.Tensor):
    """q_pool: [T, h, 64] uint8 view (row = 64 contiguous bytes) -> (token, head) strides in rows; a 4-D view
    [B, T, h, 64] -> (batch, token, head)."""
    assert q_pool.dim() in (3, 4) and q_pool.shape[-1] == 64 and q_pool.stride(-1) == 1
    assert all(st % 64 == 0 for st in q_pool.stride()[:-1])
    return tuple(st // 64 for st in q_pool.stride()[:-1])


def int4_quantize(src: torch.Tensor, q_pool: torch.Tensor, sz_pool: torch.Tensor, dst_row0: int):
    """src [S, h, 128] fp16/bf16 view -> rows dst_row0.. of q_pool [T, h, 64] u8 / sz_pool [T, h, 2] f16."""
    lib = load_library()
    if src.shape[0] == 0 or src.shape[1] == 0:
        return
    if src.dtype not in (torch.float16, torch.bfloat16):
        raise DuoHipError(f"int4_quantize: fp16 or bf16 input, got {src.dtype}")
    _require_gpu(src, "src", src.dtype)
    _require_gpu(q_pool, "q_pool", torch.uint8)
    _require_gpu(sz_pool, "sz_pool", torch.float16)
    ts, hs = _pool_row_strides(q_pool)
    assert sz_pool.stride(0) == 2 * ts and sz_pool.stride(1) == 2 * hs and sz_pool.stride(2) == 1
    assert src.stride(2) == 1 and dst_row0 + src.shape[0] <= q_pool.shape[0]
    _check(lib.duo_int4_quantize(src.data_ptr(), int(src.dtype == torch.bfloat16), src.stride(0), src.stride(1),
                                 q_pool.data_ptr(), sz_pool.data_ptr(), ts, hs, src.shape[1], src.shape[0],
                                 int(dst_row0), src.shape[2], _stream_ptr()), "duo_int4_quantize")


def int4_quantize_batched(src: torch.Tensor, q_pool: torch.Tensor, sz_pool: torch.Tensor, dst_row0: int):
    """src [B, S, h, 128] fp16/bf16 -> rows dst_row0.. of q_pool [B, T, h, 64] u8 / sz_pool [B, T, h, 2] f16, every batch
    row in one launch."""
    lib = load_library()
    if src.shape[0] == 0 or src.shape[1] == 0 or src.shape[2] == 0:
        return
    if src.dtype not in (torch.float16, torch.bfloat16):
        raise DuoHipError(f"int4_quantize: fp16 or bf16 input, got {src.dtype}")
    _require_gpu(src, "src", src.dtype)
    _require_gpu(q_pool, "q_pool", torch.uint8)
    _require_gpu(sz_pool, "sz_pool", torch.float16)
    bs, ts, hs = _pool_row_strides(q_pool)
    assert sz_pool.stride() == (2 * bs, 2 * ts, 2 * hs, 1)
    assert src.dim() == 4 and src.stride(3) == 1 and dst_row0 + src.shape[1] <= q_pool.shape[1] and q_pool.shape[0] == src.shape[0]
    _check(lib.duo_int4_quantize_batched(src.data_ptr(), int(src.dtype == torch.bfloat16), src.stride(0), src.stride(1),
                                         src.stride(2), q_pool.data_ptr(), sz_pool.data_ptr(), bs, ts, hs, src.shape[0],
                                         src.shape[2], src.shape[1], int(dst_row0), src.shape[3], _stream_ptr()),
           "duo_int4_quantize_batched")


def int4_dequantize_batched(q_pool: torch.Tensor, sz_pool: torch.Tensor, n_tokens: int, out: torch.Tensor,
                            fused: bool = False) -> torch.Tensor:
    """rows [0, n_tokens) of every batch row of the pool [B, T, h, 64] -> out viewed [B, n_tokens, h, 128] fp16."""
    lib = load_library()
    B, h = q_pool.shape[0], q_pool.shape[2]
    per = n_tokens * h * HEAD_DIM
    res = out[: B * per].view(B, n_tokens, h, HEAD_DIM)
    if n_tokens == 0 or h == 0 or B == 0:
        return res
    _require_gpu(q_pool, "q_pool", torch.uint8)
    _require_gpu(out, "out", torch.float16)
    bs, ts, hs = _pool_row_strides(q_pool)
    _check(lib.duo_int4_dequantize_batched_f16(q_pool.data_ptr(), sz_pool.data_ptr(), bs, ts, hs, res.data_ptr(), per, B, h,
                                               int(n_tokens), HEAD_DIM, int(bool(fused)), _stream_ptr()),
           "duo_int4_dequantize_batched_f16")
    return res


def int4_stream_compress_batched(kq, ksz, vq, vsz, length: int, sink: int, recent: int) -> int:
    """pools [B, T, h, ...]: keep sink + recent of `length` rows in every batch row, one launch"""
    lib = load_library()
    new_len = c_int32(0)
    B, h = kq.shape[0], kq.shape[2]
    bs, ts, hs = _pool_row_strides(kq) if (h and B) else (0, 0, 0)
    live = bool(h and B)
    _check(lib.duo_int4_stream_compress_batched(kq.data_ptr() if live else None, ksz.data_ptr() if live else None,
                                                vq.data_ptr() if live else None, vsz.data_ptr() if live else None, bs, ts,
                                                hs, B, h, int(length), int(sink), int(recent), byref(new_len),
                                                _stream_ptr()), "duo_int4_stream_compress_batched")
    return int(new_len.value)


def int4_dequantize(q_pool: torch.Tensor, sz_pool: torch.Tensor, n_tokens: int, out: torch.Tensor,
                    fused: bool = False) -> torch.Tensor:
    """rows [0, n_tokens) of the pool -> out[: n_tokens*h*128] viewed [n_tokens, h, 128] fp16.  ``fused``: q*s + z as
    one fma (one rounding) instead of the source's hmul then hadd (two) — see ``duo_int4_dequantize_f16``."""
    lib = load_library()
    h = q_pool.shape[1]
    res = out[: n_tokens * h * HEAD_DIM].view(n_tokens, h, HEAD_DIM)
    if n_tokens == 0 or h == 0:
        return res
    _require_gpu(q_pool, "q_pool", torch.uint8)
    _require_gpu(out, "out", torch.float16)
    ts, hs = _pool_row_strides(q_pool)
    _check(lib.duo_int4_dequantize_f16(q_pool.data_ptr(), sz_pool.data_ptr(), ts, hs, res.data_ptr(), h,
                                       int(n_tokens), HEAD_DIM, int(bool(fused)), _stream_ptr()), "duo_int4_dequantize_f16")
    return res


def int4_stream_compress(kq, ksz, vq, vsz, length: int, sink: int, recent: int) -> int:
    lib = load_library()
    new_len = c_int32(0)
    h = kq.shape[1]
    ts, hs = _pool_row_strides(kq) if h else (0, 0)
    _check(lib.duo_int4_stream_compress(kq.data_ptr() if h else None, ksz.data_ptr() if h else None,
                                        vq.data_ptr() if h else None, vsz.data_ptr() if h else None, ts, hs, h,
                                        int(length), int(sink), int(recent), byref(new_len), _stream_ptr()),
           "duo_int4_stream_compress")
    return int(new_len.value)


def make_int4_pool(kq, ksz, vq, vsz, length: int, q_head_offset: int) -> Optional[Int4Pool]:
    """kq/vq [T, h, 64] u8, ksz/vsz [T, h, 2] f16 views of one head class (or [B, T, h, ...] for the batched decode)."""
    if kq is None or kq.shape[-2] == 0 or length <= 0:
        return None
    p = Int4Pool()
    p.k_q, p.v_q, p.k_sz, p.v_sz = kq.data_ptr(), vq.data_ptr(), ksz.data_ptr(), vsz.data_ptr()
    strides = _pool_row_strides(kq)
    assert _pool_row_strides(vq) == strides
    p.token_stride_rows, p.head_stride_rows = strides[-2], strides[-1]
    p.batch_stride_rows = strides[0] if kq.dim() == 4 else 0
    p.len, p.n_kv_heads, p.q_head_offset = int(length), kq.shape[-2], int(q_head_offset)
    return p


def _int4_mode(fused) -> int:
    """``fused`` of the INT4 decode entry points: False / 0 = the reference's hmul-then-hadd values, True / 1 = its fma
    values (both: dequantise in registers, bit-pinned), 2 / "folded" = no per-element dequantisation in tiles whose rows
    are tame — scale and zero are applied to the score tile and to P (``duo_int4_decode_fold_kernel``: the values n s + z
    un-rounded); tiles with outlier rows dequantise element by element in the form 0 (mode 2) or 1 (mode 3)"""
    if fused == "folded":
        return 2
    if fused in (2, 3) and not isinstance(fused, bool):
        return int(fused)           # 3: folded, with the fma form in the tiles that take the exact body
    return int(bool(fused))


def attn_decode_int4(q: torch.Tensor, out: torch.Tensor, group: int, full: Optional[Int4Pool],
                     stream: Optional[Int4Pool], scale: float, fused=False):
    """q, out [Hq, 128] fp16: decode attention over the packed pools; ``fused``: see ``_int4_mode``."""
    lib = load_library()
    _require_gpu(q, "q", torch.float16)
    _require_gpu(out, "out", torch.float16)
    assert q.dim() == 2 and q.stride(1) == 1 and out.shape == q.shape
    ws = decode_workspace(q.device, q.shape[0])
    _check(lib.duo_attn_decode_int4_f16(q.data_ptr(), q.stride(0), out.data_ptr(), out.stride(0), int(group),
                                        byref(full) if full is not None else None,
                                        byref(stream) if stream is not None else None, float(scale), q.shape[1],
                                        _int4_mode(fused), ws.data_ptr(), ws.numel() * 4, _stream_ptr()),
           "duo_attn_decode_int4_f16")


def attn_decode_int4_batched(q: torch.Tensor, out: torch.Tensor, group: int, full: Optional[Int4Pool],
                             stream: Optional[Int4Pool], scale: float, fused=False):
    """q, out [B, Hq, 128] fp16; pools built from [B, T, h, ...] views: every batch row in one launch pair"""
    lib = load_library()
    _require_gpu(q, "q", torch.float16)
    _require_gpu(out, "out", torch.float16)
    assert q.dim() == 3 and q.stride(2) == 1 and out.shape == q.shape
    ws = decode_workspace(q.device, q.shape[1])
    _check(lib.duo_attn_decode_int4_batched_f16(q.data_ptr(), q.stride(0), q.stride(1), out.data_ptr(), out.stride(0),
                                                out.stride(1), q.shape[0], int(group),
                                                byref(full) if full is not None else None,
                                                byref(stream) if stream is not None else None, float(scale), q.shape[2],
                                                _int4_mode(fused), ws.data_ptr(), ws.numel() * 4, _stream_ptr()),
           "duo_attn_decode_int4_batched_f16")


def set_debug_flags(flags: int):
    load_library().duo_set_debug_flags(int(flags) | _ENV_DEBUG_FLAGS)


def last_prefill_plan():
    """(key-range pieces of the retrieval class, of the streaming class, estimated microseconds, estimated microseconds
    unsplit) of this thread's last prefill launch — the launcher's plan (csrc/duo_prefill.hip), for tests and probes."""
    out = (ctypes.c_double * 4)()
    load_library().duo_debug_prefill_last_plan(out)
    return int(out[0]), int(out[1]), float(out[2]), float(out[3])


def prefill_plan(nkv0, nkv1, group, n_tokens, lenA0, lenB0, lenA1, lenB1, max_parts=2048, xmap=3, force=0, with_blocks=False):
    """host-only: the launcher's plan for a prefill launch shape (``duo_debug_prefill_plan``): dict with k0, k1, blocks,
    partials, blocks0, est_us, est_unsplit_us and — ``with_blocks`` — an int32 array [blocks, 6] of
    (class, q tile | -1, kv head, q head of the group, piece, partial slot) per block id"""
    import numpy as np

    lib = load_library()
    shape = (c_int32 * 10)(nkv0, nkv1, group, n_tokens, lenA0, lenB0, lenA1, lenB1, max_parts, xmap)
    out, est = (c_int32 * 5)(), (ctypes.c_double * 2)()
    n = lib.duo_debug_prefill_plan(shape, force, out, est, None, 0)
    if n < 0:
        raise DuoHipError(f"duo_debug_prefill_plan: {n}")
    res = {"k0": out[0], "k1": out[1], "blocks": out[2], "partials": out[3], "blocks0": out[4], "est_us": est[0],
           "est_unsplit_us": est[1]}
    if with_blocks:
        arr = np.zeros((max(n, 1), 6), dtype=np.int32)
        lib.duo_debug_prefill_plan(shape, force, out, est, arr.ctypes.data_as(POINTER(c_int32)), n)
        res["map"] = arr[:n]
    return res
