"""rms error of the HIP prefill kernel against the exact-P oracle, relative to the reference arithmetic's own (oracle round_p=True)
on the same inputs, per merge form / split setting (tests/helpers.attn_close holds the ratio to 1.03 + 3 / sqrt(n))."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "duo-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import test_hip_kernels_gpu as T  # noqa: E402
from duo_attn import _hip  # noqa: E402

CASES = [(64, 4, 2, 2, 5, 384), (100, 4, 1, 3, 1000, 384), (300, 4, 0, 2, 0, 384), (700, 4, 3, 5, 3000, 330), (1000, 4, 1, 1, 1000, 384)]
FORMS = [("in-kernel", 0), ("launch pair", 1 << 20), ("no split", 256), ("w8x32 no split", 256 | 128), ("forced 2,2 pair", (2 << 12) | (2 << 16) | (1 << 20)),
         ("forced 2,2 in-kernel", (2 << 12) | (2 << 16))]
for case in CASES:
    for name, flags in FORMS:
        _hip.set_debug_flags(flags)
        try:
            out, ref, bud = T._attention_case(*case, True, seed=hash(case) % 1000)
            torch.cuda.synchronize()
            plan = _hip.last_prefill_plan()
        finally:
            _hip.set_debug_flags(0)
        o, r, rr = out.float().cpu(), ref.float(), bud.ref_rounded.float()
        e, en = (o - r).pow(2).mean().sqrt(), (rr - r).pow(2).mean().sqrt()
        print(f"{str(case):34s} {name:22s} plan {plan[:2]}  rms err {e:.4e}  reference arithmetic {en:.4e}  ratio {e / en:.3f}", flush=True)
