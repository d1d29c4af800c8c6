import sys, os
sys.path.insert(0, "duo-attention_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from duo_attn import _hip
from oracle.int4_oracle import quantize_int4_ref, dequantize_int4_ref
import test_int4 as T
DEV = "cuda:0"
group, nf, ns, n_full, n_stream = 4, 2, 6, 300, 385
for mode in ("tiny", "big", "both"):
    g = torch.Generator().manual_seed(n_full + n_stream)
    Hq = (nf + ns) * group
    q = torch.randn(Hq, 128, generator=g).to(torch.float16)
    ref = torch.empty(Hq, 128)
    pools = []
    for n_h, TT, off in ((nf, n_full, 0), (ns, n_stream, nf * group)):
        k = torch.randn(TT, n_h, 128, generator=g)
        v = torch.randn(TT, n_h, 128, generator=g)
        for x, big in ((k, 300.0), (v, 800.0)):
            f = torch.ones(TT, n_h, 1)
            u = torch.rand(TT, n_h, 1, generator=g)
            if mode in ("tiny", "both"): f[u < 0.1] = 1e-3
            if mode in ("big", "both"): f[u > 0.995] = big
            x *= f
        k, v = k.to(torch.float16), v.to(torch.float16)
        kq, ksz = T._pools(TT + 3, n_h)
        vq, vsz = T._pools(TT + 3, n_h)
        _hip.int4_quantize(k.to(DEV), kq, ksz, 0)
        _hip.int4_quantize(v.to(DEV), vq, vsz, 0)
        pools.append(_hip.make_int4_pool(kq, ksz, vq, vsz, TT, off))
        kd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(k.float().numpy())).astype(np.float32))
        vd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(v.float().numpy())).astype(np.float32))
        ref[off:off + n_h * group] = T._ref_attention(q.float()[off:off + n_h * group], kd, vd, group)
        pools[-1]._keep = (kq, ksz, vq, vsz)
    for flags in (0, 16):
        _hip.set_debug_flags(flags)
        out = torch.full((Hq, 128), float("nan"), dtype=torch.float16, device=DEV)
        _hip.attn_decode_int4(q.to(DEV), out, group, pools[0], pools[1], 128 ** -0.5)
        o = out.float().cpu()
        err = (o - ref).abs()
        tol = 1e-3 * ref.abs() + 2.0 ** -10 * ref.abs() + 1e-3 * ref.pow(2).mean().sqrt()
        bad = err > tol
        i = err.argmax()
        print(mode, "flags", flags, "finite", bool(torch.isfinite(o).all()), "nbad", int(bad.sum()), "maxerr", float(err.max()),
              "ref@", float(ref.flatten()[i]), "rms", float(ref.pow(2).mean().sqrt()), "bad heads", sorted(set(torch.nonzero(bad)[:, 0].tolist())))
_hip.set_debug_flags(0)
