"""element-level view of a tests/fuzz_token_linear.py case:  python tests/fuzz_linear_detail.py "<dict>" """
import ast
import sys

import torch

import fuzz_token_linear as F
from duo_attn import _hip
from oracle.duo_oracle import token_linear_ref

c = ast.literal_eval(sys.argv[1])
rows, n_in, sizes, pro, pad = c["rows"], c["n_in"], c["sizes"], c["pro"], c["pad"]
g = torch.Generator().manual_seed(c["seed"])
x = F._rand((rows, n_in), g, c["scale"])
x2 = F._rand((rows, n_in), g) if pro == "silu" else None
blocks = [(F._rand((n, n_in), g, scale=n_in ** -0.5), F._rand((n,), g) if c["bias"] else None) for n in sizes]
norm = (F._rand((n_in,), g).abs() + 0.5, 1e-5) if pro.startswith("norm") else None
res = F._rand((rows, sum(sizes)), g) if c["residual"] else None
ref, pre = token_linear_ref(x, blocks, norm=norm, x2=x2, residual=res, exact=True, norm_hf=pro == "norm_hf")
wd = [(w.to("cuda:0"), None if b is None else b.to("cuda:0")) for w, b in blocks]
for trial in range(3):
    y = _hip.token_linear(x.to("cuda:0"), wd, norm=None if norm is None else (norm[0].to("cuda:0"), norm[1]),
                          x2=None if x2 is None else x2.to("cuda:0"), residual=None if res is None else res.to("cuda:0"),
                          norm_hf=pro == "norm_hf").float().cpu()
    tol = (2.0 ** -7) * pre.float().abs() + 1e-4
    if res is not None:
        tol = tol + (2.0 ** -7) * (pre.float() + res.float()).abs()
    err = (y - ref.float()).abs()
    bad = (err > tol).nonzero()
    print("trial", trial, "bad", bad.tolist())
    for r, col in bad.tolist():
        print(f"   row {r} col {col}: y {y[r, col]:.6f} ref {ref[r, col].float():.6f} pre {pre[r, col]:.8f} res {None if res is None else float(res[r, col])} bias {None if blocks[0][1] is None else float(torch.cat([b for _, b in blocks])[col])}")
        # the same output with the device-normalised x fed to the oracle
if norm is not None:
    xn_dev = _hip.rmsnorm(x.to("cuda:0"), norm[0].to("cuda:0"), norm[1]).cpu()
    from oracle.duo_oracle import rmsnorm_ref
    xn_ref = rmsnorm_ref(x, norm[0], norm[1])
    d = (xn_dev.float() - xn_ref.float()).abs()
    print("normalised rows: elements that differ device vs oracle:", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()))
    ref2, pre2 = token_linear_ref(xn_dev, blocks, norm=None, x2=None, residual=res, exact=True)
    for r, col in bad.tolist():
        print(f"   with device-normalised x: ref {ref2[r, col].float():.6f} pre {pre2[r, col]:.8f}")
