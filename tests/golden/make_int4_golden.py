"""Golden vectors for the INT4 KV pools, produced by the REFERENCE'S OWN KERNELS.

    python oracle/build_ref.py                       # build container (needs /root/reference): oracle/_ref/*.so
    gpurun -- python tests/golden/make_int4_golden.py     # GPU box: writes gpurun_out/int4_ref.npz
    cp gpurun_out/int4_ref.npz tests/golden/int4_ref.npz  # commit

Runs `quantize_int4_with_zero_point_per_group` / `dequantize_int4_with_zero_point_per_group`
(/root/reference/demo/quantize_int4.cu:44-71,146-178, bound exactly as demo/int4_kv.py:59-112 calls them) from
the three hipcc builds of that file described in oracle/build_ref.py and records inputs + outputs.  Needs a
GPU (the kernels are device code); reads nothing from /root/reference at run time.

File layout (all arrays little-endian; fp16 stored as uint16 bit patterns so numpy versions cannot disagree):
    x|<case>                      uint16 [S, H, 128]   fp16 input rows
    strided|<case>                uint8                1 if the tensor handed to the kernel was a strided view
    q|<case>|<variant>            uint8  [S, H, 64]    packed codes
    s|<case>|<variant>, z|...     uint16 [S, H]        fp16 scale / zero point
    dq|<case>|<variant>           uint16 [S, H, 128]   dequantised with the SAME variant's kernel
    rawq, raws, rawz              random packed bytes / fp16 scale / fp16 zero (dequantise-only case)
    rawdq|<variant>               uint16 [N, 128]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.build_ref import VARIANTS, load_ref  # noqa: E402

DEV = "cuda:0"
GROUP = 128


def cases():
    g = torch.Generator().manual_seed(20250925)
    out = {}
    out["randn"] = (torch.randn(48, 3, 128, generator=g) * 2).half()
    out["bf16vals"] = (torch.randn(32, 2, 128, generator=g) * 3).bfloat16().half()   # exactly representable
    out["small"] = (torch.randn(16, 2, 128, generator=g) * 1e-3).half()
    out["large"] = (torch.randn(16, 2, 128, generator=g) * 3000).half()
    out["offset"] = (torch.randn(16, 2, 128, generator=g) * 0.05 + 7.0).half()        # narrow range far from 0
    sp = torch.zeros(24, 1, 128)
    sp[0] = 0.75                                   # constant row: scale = 1e-8 -> fp16 0
    sp[1] = 0.0
    sp[2, 0, :64], sp[2, 0, 64:] = 10.0, -3.0      # two-level row: codes 0 / 15 only
    sp[3, 0, ::2], sp[3, 0, 1::2] = 65504.0, -65504.0   # widest fp16 range
    sp[4, 0, :5] = 65504.0                         # [0, max]
    sp[5] = 1.0
    sp[5, 0, 7] = 1.0 + 2.0 ** -10                 # one-ulp range
    sp[6, 0] = torch.arange(128) * 2.0 ** -24      # fp16 subnormals
    sp[7, 0] = torch.randn(128, generator=g) * 0.1
    sp[7, 0, 100] = 40.0                           # one outlier
    sp[8, 0] = torch.arange(128).float()           # integer ramp, scale 127/15
    sp[9, 0] = (torch.arange(128) % 16).float()    # scale 1 + 1e-8: exact codes
    sp[10, 0] = (torch.arange(128) % 16).float() * 0.5 + 0.25   # values on half-code boundaries
    sp[10, 0, 0], sp[10, 0, 1] = 0.0, 7.5
    sp[11, 0] = -torch.rand(128, generator=g) * 5   # negative only
    sp[12, 0] = torch.linspace(-1, 1, 128)
    sp[13, 0] = torch.linspace(0, 15, 128) * 0.3    # many near-tie quotients
    sp[14, 0] = torch.linspace(-8, 8, 128).round()  # integers, scale 16/15
    sp[15, 0, ::3] = -0.0
    sp[15, 0, 1::3] = 2.0 ** -14
    for r in range(16, 24):                         # quotients engineered onto k + 0.5: x = zero + (k + .5) * step
        step = float(2.0 ** (r - 20))
        ks = torch.arange(128) % 15
        sp[r, 0] = (ks + 0.5) * step
        sp[r, 0, 0], sp[r, 0, 1] = 0.0, 15 * step
    out["special"] = sp.half()
    return out


def run_quant(mod, x, strided):
    S, H, D = x.shape
    if strided:   # the kernel takes strides(0..2) (quantize_int4.cu:171-173): hand it a head slice of a wider tensor
        wide = torch.zeros(1, S, H + 3, D, dtype=torch.float16, device=DEV)
        wide[:, :, 2:2 + H] = x.to(DEV)
        t = wide[:, :, 2:2 + H]
    else:
        t = x.to(DEV).view(1, S, H, D).contiguous()
    q = torch.zeros(1, S, H, D // 2, dtype=torch.uint8, device=DEV)
    s = torch.zeros(1, S, H, 1, dtype=torch.float16, device=DEV)
    z = torch.zeros(1, S, H, 1, dtype=torch.float16, device=DEV)
    mod.quantize_int4_with_zero_point_per_group(t, q, s, z, GROUP)
    torch.cuda.synchronize()
    return q[0], s[0, ..., 0], z[0, ..., 0]


def run_dequant(mod, q, s, z):
    qq = q.reshape(-1, GROUP // 2).contiguous()
    N = qq.shape[0]
    buf = torch.zeros(N * GROUP, dtype=torch.float16, device=DEV)
    mod.dequantize_int4_with_zero_point_per_group(qq, s.reshape(-1).contiguous(), z.reshape(-1).contiguous(),
                                                   GROUP, buf, N)
    torch.cuda.synchronize()
    return buf.view(*q.shape[:-1], GROUP)


def u16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def main():
    assert torch.cuda.is_available(), "needs a GPU: the reference kernels are device code"
    mods = {v: load_ref(v) for v in VARIANTS}
    arrays = {}
    cs = cases()
    for ci, (name, x) in enumerate(cs.items()):
        strided = ci % 2 == 1
        arrays[f"x|{name}"] = u16(x)
        arrays[f"strided|{name}"] = np.uint8(strided)
        for v, mod in mods.items():
            q, s, z = run_quant(mod, x, strided)
            arrays[f"q|{name}|{v}"] = q.cpu().numpy()
            arrays[f"s|{name}|{v}"] = u16(s)
            arrays[f"z|{name}|{v}"] = u16(z)
            arrays[f"dq|{name}|{v}"] = u16(run_dequant(mod, q, s, z))
    # dequantise-only: arbitrary packed bytes and (scale, zero) pairs, not tied to any quantiser
    g = torch.Generator().manual_seed(7)
    N = 384
    rawq = torch.randint(0, 256, (N, GROUP // 2), generator=g, dtype=torch.uint8)
    mag = torch.exp(torch.rand(N, generator=g) * 16 - 11)                       # e^-11 .. e^5
    raws = (mag * torch.where(torch.rand(N, generator=g) < 0.1, -1.0, 1.0)).half()
    rawz = (torch.randn(N, generator=g) * torch.exp(torch.rand(N, generator=g) * 10 - 6)).half()
    raws[:8] = torch.tensor([0.0, 1.0, 2.0 ** -24, 2.0 ** -14, 4368.0, 0.33325, 1e-3, 60.0]).half()
    rawz[:8] = torch.tensor([1.5, 0.0, 0.0, -2.0 ** -14, -65504.0, -2.5, 0.1, -450.0]).half()
    arrays["rawq"], arrays["raws"], arrays["rawz"] = rawq.numpy(), u16(raws), u16(rawz)
    for v, mod in mods.items():
        arrays[f"rawdq|{v}"] = u16(run_dequant(mod, rawq.to(DEV), raws.to(DEV), rawz.to(DEV)))

    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "int4_ref.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {len(arrays)} arrays, {os.path.getsize(path)} bytes")

    # summary of how the three builds differ (the numbers DESIGN.md quotes)
    for name in cs:
        for v in ("default", "fast"):
            dq = (arrays[f"q|{name}|{v}"] != arrays[f"q|{name}|nocontract"]).sum()
            ds = (arrays[f"s|{name}|{v}"] != arrays[f"s|{name}|nocontract"]).sum()
            dd = (arrays[f"dq|{name}|{v}"] != arrays[f"dq|{name}|nocontract"]).sum()
            print(f"  {name:9s} {v:8s} vs nocontract: packed bytes differ {dq}/{arrays[f'q|{name}|{v}'].size}, "
                  f"scales {ds}, dequantised halves {dd}/{arrays[f'dq|{name}|{v}'].size}")
    for v in ("default", "fast"):
        print(f"  raw dequant {v} vs nocontract: {(arrays[f'rawdq|{v}'] != arrays['rawdq|nocontract']).sum()}/{N * GROUP}")


if __name__ == "__main__":
    main()
