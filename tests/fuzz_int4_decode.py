"""Randomised differential run of the fused INT4 decode attention (duo_attn_decode_int4_f16) against the oracle: random GQA
group, head counts per class, pool lengths (1 ... 60 000 rows), batch rows (the batched entry points), head-major / token-major pools, rows with extreme scales, the
default (dequantising) kernel and the opt-in folded one.  Same check as tests/test_int4.py::test_fused_int4_decode.

    python tests/fuzz_int4_decode.py --seconds 120 [--seed 1] [--folded]

Reference restated by the oracle: demo/quantize_int4.cu:9-178 (values), demo/int4_kv.py:373-436 (attention over them)."""
import argparse
import math
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "duo-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.int4_oracle import dequantize_int4_ref, quantize_int4_ref  # noqa: E402

DEV = "cuda:0"


def _pools(T, h, head_major, B=0):
    """[T, h, ...] views (B == 0) or [B, T, h, ...] views for the batched entry points"""
    lead = (B,) if B else ()
    if head_major:
        q = torch.zeros(*lead, h, T, 64, dtype=torch.uint8, device=DEV).transpose(-3, -2)
        sz = torch.zeros(*lead, h, T, 2, dtype=torch.float16, device=DEV).transpose(-3, -2)
    else:
        q = torch.zeros(*lead, T, h, 64, dtype=torch.uint8, device=DEV)
        sz = torch.zeros(*lead, T, h, 2, dtype=torch.float16, device=DEV)
    return q, sz


def _ref_attention(q, kd, vd, group, budget):
    Hq = q.shape[0]
    out = torch.empty(Hq, 128)
    for hq in range(Hq):
        k, v = kd[:, hq // group], vd[:, hq // group]
        p = torch.softmax((k @ q[hq]) / (128 ** 0.5), 0)
        out[hq] = p @ v
        budget[hq] = p @ v.abs()
    return out


def draw_case(rng, folded=False):
    group = rng.choice([1, 2, 3, 4, 4, 4, 5, 7, 8, 16])
    nf, ns = rng.randint(0, 4), rng.randint(0, 4)
    if nf + ns == 0:
        nf = 1
    lens = lambda hi: max(1, int(math.exp(rng.uniform(0, math.log(hi)))))
    return dict(group=group, nf=nf, ns=ns, n_full=lens(60000), n_stream=lens(700), odd_rows=rng.random() < 0.4,
                mode=rng.choice([0, 0, 2]) if folded else 0, B=rng.choice([1, 1, 2, 3]), head_major=rng.random() < 0.75, pad=rng.randint(0, 5), scale=rng.choice([0.3, 1.0, 1.0, 2.0]),
                seed=rng.randint(0, 2 ** 31 - 1))


def run_case(c):
    from duo_attn import _hip

    g = torch.Generator().manual_seed(c["seed"])
    group, nf, ns, B = c["group"], c["nf"], c["ns"], c.get("B", 1)
    Hq = (nf + ns) * group
    q = (torch.randn(B, Hq, 128, generator=g) * c["scale"]).to(torch.float16)
    ref, bud = torch.empty(B, Hq, 128), torch.empty(B, Hq, 128)
    pools = []
    for n_h, T, off in ((nf, c["n_full"], 0), (ns, c["n_stream"], nf * group)):
        if n_h == 0:
            pools.append(None)
            continue
        k = torch.randn(B, T, n_h, 128, generator=g) * c["scale"]
        v = torch.randn(B, T, n_h, 128, generator=g) * c["scale"]
        if c["odd_rows"]:
            for x, big in ((k, 300.0), (v, 800.0)):
                f = torch.ones(B, T, n_h, 1)
                u = torch.rand(B, T, n_h, 1, generator=g)
                f[u < 0.1] = 1e-3
                f[u > 0.995] = big
                x *= f
        k, v = k.to(torch.float16), v.to(torch.float16)
        kq, ksz = _pools(T + c["pad"], n_h, c["head_major"], B if B > 1 else 0)
        vq, vsz = _pools(T + c["pad"], n_h, c["head_major"], B if B > 1 else 0)
        if B > 1:
            _hip.int4_quantize_batched(k.to(DEV), kq, ksz, 0)
            _hip.int4_quantize_batched(v.to(DEV), vq, vsz, 0)
        else:
            _hip.int4_quantize(k[0].to(DEV), kq, ksz, 0)
            _hip.int4_quantize(v[0].to(DEV), vq, vsz, 0)
        pools.append(_hip.make_int4_pool(kq, ksz, vq, vsz, T, off))
        pools[-1]._keep = (kq, ksz, vq, vsz)
        for b in range(B):
            kd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(k[b].float().numpy())).astype(np.float32))
            vd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(v[b].float().numpy())).astype(np.float32))
            ref[b, off:off + n_h * group] = _ref_attention(q[b].float()[off:off + n_h * group], kd, vd, group,
                                                           bud[b, off:off + n_h * group])
    out = torch.full((B, Hq, 128), float("nan"), dtype=torch.float16, device=DEV)
    if B > 1:
        _hip.attn_decode_int4_batched(q.to(DEV), out, group, pools[0], pools[1], 128 ** -0.5, fused=c["mode"])
    else:
        _hip.attn_decode_int4(q[0].to(DEV), out[0], group, pools[0], pools[1], 128 ** -0.5, fused=c["mode"])
    o = out.float().cpu()
    assert torch.isfinite(o).all(), "non-finite output"
    err = (o - ref).abs()
    rms = ref.pow(2).mean(dim=(1, 2), keepdim=True).sqrt()
    tol = 1e-3 * ref.abs() + 2.0 ** -10 * ref.abs() + 2.0 ** -10 * bud + 1e-3 * rms
    bad = err > tol
    assert not bad.any(), (f"{int(bad.sum())}/{bad.numel()} out of tolerance (rows {sorted(set(bad.nonzero()[:, 0].tolist()))}), "
                           f"worst err/tol {float((err / tol).max()):.2f}, max err {err.max():.3e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--folded", action="store_true",
                    help="also draw the OPT-IN folded kernel (fused = 2): it is known to leave this bar where q and the data are "
                         "larger than N(0, 1) - up to 4.6x at 2 sigma, profiles/r4_int4_fold.md - so its failures are a measurement")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        c = draw_case(rng, a.folded)
        n += 1
        try:
            run_case(c)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("FAIL", c, "\n    ", f"{type(e).__name__}: {str(e)[:400]}", flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s, {bad} failed (seed {a.seed})")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
