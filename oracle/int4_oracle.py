"""CPU oracle for the INT4 KV pools (BASELINE config 5).

TEST INFRASTRUCTURE ONLY (same rules as duo_oracle.py).  numpy restatement, bit level, of the only
native code in the reference, demo/quantize_int4.cu:

  quantize   (:73-144)  per (token, head) row of `group_size` = 128 values: fp32 min / max,
             scale = (max - min) / 15 + 1e-8,  zero = min,
             q = clamp(roundf((x - zero) / scale), 0, 15)   [roundf: half away from zero]
             packed byte i = (q[2i] << 4) | q[2i+1]          [even element -> HIGH nibble]
             scale, zero stored as fp16 (the kernel quantises with the fp32 scale, stores the rounded one)
  dequantize (:9-42)    out = hadd(hmul(half(q), scale), zero) — fp16 arithmetic, two roundings.

PINNED against the reference's own kernels: oracle/build_ref.py compiles that .cu (hipified by
torch.utils.cpp_extension, as the reference loads it) for gfx950 in three flag variants, tests/golden/
make_int4_golden.py ran them on an MI355X, and tests/test_int4_golden.py compares this file with the
recorded outputs (tests/golden/int4_ref.npz):

  * `nocontract` build (the source as written: IEEE division, hmul then hadd) == this oracle's defaults,
    bit for bit, every case.  This is also what the product's HIP kernels implement.
  * `default` hipcc build: clang's default -ffp-contract=fast fuses hmul+hadd into v_pk_fma_f16 (one
    rounding) == this oracle with ``fused=True``, bit for bit; quantisation identical to `nocontract`.
  * `fast` build (-ffast-math, the hipcc spelling of the reference's nvcc --use_fast_math): the division
    becomes x * v_rcp_f32(scale), a hardware approximation no CPU restatement can reproduce bit-exactly;
    the test bounds the disagreement (codes differ by at most 1, only where the exact quotient is within
    rounding distance of k + 0.5) and records how many do.
What stays out of reach: the NVIDIA binary itself (nvcc's div.approx.ftz and whatever ptxas does with
mul.f16 + add.f16) — no CUDA toolchain or device exists here; on any platform the flags, not the source,
pick between the variants above.
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


def dequantize_int4_ref(packed: np.ndarray, scale: np.ndarray, zero: np.ndarray, fused: bool = False) -> np.ndarray:
    """packed [..., 64] uint8, scale/zero [...] float16 -> [..., 128] float16.
    fused=False: the source as written, hadd(hmul(half(q), s), z) — two fp16 roundings (numpy float16
    arithmetic rounds every operation to nearest even).  fused=True: one fma.f16, what a contracting
    compiler makes of it (q*s + z is exact in float64: 4-bit x 11-bit product plus an 11-bit addend)."""
    hi = (packed >> 4).astype(np.float16)
    lo = (packed & 0x0F).astype(np.float16)
    q = np.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=np.float16)
    q[..., 0::2] = hi
    q[..., 1::2] = lo
    s = scale.astype(np.float16)[..., None]
    z = zero.astype(np.float16)[..., None]
    if fused:
        with np.errstate(over="ignore", invalid="ignore"):
            return (q.astype(np.float64) * s.astype(np.float64) + z.astype(np.float64)).astype(np.float16)
    with np.errstate(over="ignore", invalid="ignore"):
        return ((q * s).astype(np.float16) + z).astype(np.float16)


def dequantize_int4_torch(packed, sz):
    """Same function as dequantize_int4_ref(fused=False), in torch so that it can run where the data is
    (the 3.3M-token cfg5 test dequantises 1.7e9 values; numpy float16 would take minutes).
    packed [..., 64] uint8, sz [..., 2] float16 (scale, zero) -> [..., 128] float16.  torch's float16
    multiply and add each round to nearest even once (CPU and GPU): hmul then hadd, two roundings.
    tests/test_int4_golden.py::test_torch_restatement_equals_numpy_oracle keeps the two identical."""
    import torch

    hi = (packed >> 4).to(torch.float16)
    lo = (packed & 15).to(torch.float16)
    n = torch.stack([hi, lo], -1).reshape(*packed.shape[:-1], packed.shape[-1] * 2)
    prod = n * sz[..., 0:1]
    return prod + sz[..., 1:2]
