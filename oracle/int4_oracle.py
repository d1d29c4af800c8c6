"""CPU oracle for the INT4 KV pools (BASELINE config 5).

TEST INFRASTRUCTURE ONLY (same rules as duo_oracle.py).  numpy restatement, bit level, of the only
native code in the reference, demo/quantize_int4.cu:

  quantize   (:73-144)  per (token, head) row of `group_size` = 128 values: fp32 min / max,
             scale = (max - min) / 15 + 1e-8,  zero = min,
             q = clamp(roundf((x - zero) / scale), 0, 15)   [roundf: half away from zero]
             packed byte i = (q[2i] << 4) | q[2i+1]          [even element -> HIGH nibble]
             scale, zero stored as fp16 (the kernel quantises with the fp32 scale, stores the rounded one)
  dequantize (:9-42)    out = hadd(hmul(half(q), scale), zero) — fp16 arithmetic, two roundings.

Unpinnable residue (said here once): the reference builds that file with nvcc --use_fast_math
(demo/int4_kv.py:46-56), so its division may be the approximate __fdividef and ptxas may contract
hmul+hadd into one fma.f16; neither can be reproduced without the CUDA toolchain.  This oracle is
the source-level semantics with IEEE fp32 division and separate fp16 roundings.

PARITY UNPINNED for this file: the reference ships no vectors for the INT4 kernels and its CUDA source
cannot be built or run here (no nvcc, no CUDA device), so the oracle is a reading of the source, checked
only against itself (tests/test_int4.py: layout, error bound, constant rows) — not against reference output.
"""
import numpy as np


def roundf_ref(x: np.ndarray) -> np.ndarray:
    """C roundf: nearest, ties away from zero.  (floor(x + 0.5) is NOT it: 0.49999997f + 0.5f rounds up
    to 1.0f.)  For |x| < 2^23 the fractional part x - floor(x) is exact in fp32."""
    x = x.astype(np.float32)
    a = np.abs(x)
    f = np.floor(a)
    r = f + (a - f >= np.float32(0.5)).astype(np.float32)
    return np.copysign(r, x).astype(np.float32)


def quantize_int4_ref(x: np.ndarray):
    """x: [..., 128] float16 (or any float: converted through float32 exactly like __half2float).
    Returns (packed uint8 [..., 64], scale float16 [...], zero float16 [...])."""
    xf = x.astype(np.float32)
    mn = xf.min(axis=-1, keepdims=True)
    mx = xf.max(axis=-1, keepdims=True)
    scale = ((mx - mn) / np.float32(15.0) + np.float32(1e-8)).astype(np.float32)
    qf = ((xf - mn) / scale).astype(np.float32)
    qr = roundf_ref(qf)
    q = np.clip(qr, 0, 15).astype(np.uint8)
    packed = (q[..., 0::2] << 4) | q[..., 1::2]
    return packed.astype(np.uint8), scale[..., 0].astype(np.float16), mn[..., 0].astype(np.float16)


def dequantize_int4_ref(packed: np.ndarray, scale: np.ndarray, zero: np.ndarray) -> np.ndarray:
    """packed [..., 64] uint8, scale/zero [...] float16 -> [..., 128] float16 with the reference's two
    fp16 roundings (numpy float16 arithmetic rounds every operation to nearest even)."""
    hi = (packed >> 4).astype(np.float16)
    lo = (packed & 0x0F).astype(np.float16)
    q = np.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=np.float16)
    q[..., 0::2] = hi
    q[..., 1::2] = lo
    s = scale.astype(np.float16)[..., None]
    z = zero.astype(np.float16)[..., None]
    return ((q * s).astype(np.float16) + z).astype(np.float16)
