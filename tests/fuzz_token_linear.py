"""Randomised differential run of the token-row linears (duo_token_linear_bf16) against ``token_linear_ref``: random row
counts 1-4, feature counts (any multiple of 8 that fits the LDS staging), 1-3 weight blocks of random sizes, padded weight
rows, bias / residual, every prologue (none, RMSNorm in the flashinfer and in the HF form, SiLU * mul).  Same bar as
tests/test_token_linear_gpu.py::test_token_linear_matches_oracle (one bf16 ulp of the exact value, >= 97 % bit-equal).

    python tests/fuzz_token_linear.py --seconds 90 [--seed 1]"""
import argparse
import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "duo-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.duo_oracle import token_linear_ref  # noqa: E402

DEV = "cuda:0"


def _rand(shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def draw_case(rng):
    from duo_attn import _hip

    rows = rng.choice([1, 1, 1, 2, 3, 4])
    while True:
        n_in = 8 * rng.choice([1, rng.randint(1, 64), rng.randint(1, 640), rng.randint(1, 2048)])
        if _hip.token_linear_fits(rows, n_in):
            break
    budget = max(8, 40_000_000 // n_in)
    sizes = tuple(rng.choice([rng.randint(1, 64), rng.randint(1, 4096), rng.randint(1, min(budget, 140000))])
                  for _ in range(rng.randint(1, 3)))
    while sum(sizes) > budget:
        sizes = sizes[:-1] if len(sizes) > 1 else (budget,)
    return dict(rows=rows, n_in=n_in, sizes=sizes, bias=rng.random() < 0.4, pro=rng.choice(["none", "norm", "norm_hf", "silu"]),
                residual=rng.random() < 0.5, pad=rng.choice([0, 0, 8, 16, 264]), scale=rng.choice([0.5, 1.0, 3.0]),
                seed=rng.randint(0, 2 ** 31 - 1))


def run_case(c):
    from duo_attn import _hip

    rows, n_in, sizes, pro, pad = c["rows"], c["n_in"], c["sizes"], c["pro"], c["pad"]
    g = torch.Generator().manual_seed(c["seed"])
    x = _rand((rows, n_in), g, c["scale"])
    x2 = _rand((rows, n_in), g) if pro == "silu" else None
    blocks = [(_rand((n, n_in), g, scale=n_in ** -0.5), _rand((n,), g) if c["bias"] else None) for n in sizes]
    norm = (_rand((n_in,), g).abs() + 0.5, 1e-5) if pro.startswith("norm") else None
    res = _rand((rows, sum(sizes)), g) if c["residual"] else None
    ref, pre = token_linear_ref(x, blocks, norm=norm, x2=x2, residual=res, exact=True, norm_hf=pro == "norm_hf")
    dev = lambda t: None if t is None else t.to(DEV)
    if x2 is not None:
        both = torch.cat([x, x2], 1).to(DEV)
        xd, x2d = both[:, :n_in], both[:, n_in:]
    else:
        xd, x2d = x.to(DEV), None
    wd = []
    for (w, b), n in zip(blocks, sizes):
        wf = torch.empty(n, n_in + pad, dtype=torch.bfloat16, device=DEV)
        wf[:, :n_in].copy_(w)
        wd.append((wf[:, :n_in], dev(b)))
    y = _hip.token_linear(xd, wd, norm=None if norm is None else (norm[0].to(DEV), norm[1]), x2=x2d, residual=dev(res),
                          norm_hf=pro == "norm_hf")
    torch.cuda.synchronize()
    y, ref, pre = y.float().cpu(), ref.float(), pre.float()
    tol = (2.0 ** -7) * pre.abs() + 1e-4
    if res is not None:
        tol = tol + (2.0 ** -7) * (pre + res.float()).abs()
    if norm is not None:
        # The device's rsqrt / its order of summing the squares differ from the host's in the last fp32 bit, which flips the
        # bf16 rounding of a handful of the 10^4 normalised inputs by one ulp (measured: 2 of 46 224).  An output that is a
        # near-cancellation of the product against its bias / residual then moves by that ulp times a weight — far more than
        # one ulp of ITSELF (seen twice in 917 drawn cases): two flipped inputs at the row's largest magnitude are budgeted.
        from oracle.duo_oracle import rmsnorm_hf_ref, rmsnorm_ref

        xn = (rmsnorm_hf_ref if pro == "norm_hf" else rmsnorm_ref)(x, norm[0], norm[1]).float()
        wmax = torch.cat([w.float().abs().amax(dim=1) for w, _ in blocks])                # per output column
        tol = tol + 2 * (2.0 ** -7) * xn.abs().amax(dim=1, keepdim=True) * wmax[None, :]
    err = (y - ref).abs()
    assert torch.isfinite(y).all(), "non-finite output"
    assert (err <= tol).all(), f"{int((err > tol).sum())} of {err.numel()} beyond one ulp, worst {err.max().item():.3e}"
    same = (y == ref).float().mean().item()
    if norm is not None and same < 0.97:
        # With a norm prologue ONE flipped normalised input (above) shifts EVERY output by (one ulp of that input) x (its
        # weight): some 6-25 % of an output ulp, so that share of the outputs lands on the other side of a rounding boundary —
        # all still inside the one-ulp bar just checked (round 6, seed 6107: 0.911 bit-equal; the host's rsqrt moved by one
        # fp32 ulp reproduces exactly that).  The bit-equal share is therefore taken against the oracle with its rsqrt moved
        # by up to two fp32 ulps either way, the best per row.
        xf = x.float()
        r0 = torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + norm[1])
        best = torch.zeros(rows)
        for k in range(-4, 5):
            r = r0 * (1.0 + k * 2.0 ** -24)
            if pro == "norm_hf":
                xn_k = norm[0] * (xf * r).to(x.dtype)
            else:
                xn_k = (xf * r * norm[0].float()).to(x.dtype)
            yk = torch.cat([xn_k.double() @ w.double().t() + (0 if b is None else b.double()) for w, b in blocks], -1).float().to(x.dtype)
            if res is not None:
                yk = (res.float() + yk.float()).to(x.dtype)
            best = torch.maximum(best, (y == yk.float()).float().mean(dim=1))
        same = best.mean().item()
    assert same >= 0.97 or y.numel() < 200, f"only {same:.4f} of the elements bit-equal to the oracle"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", default=None, help="re-run ONE drawn case (the dict a FAIL line printed) and exit")
    a = ap.parse_args()
    if a.case:
        import ast
        run_case(ast.literal_eval(a.case))
        print("case passed")
        return
    rng = random.Random(a.seed)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        c = draw_case(rng)
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
