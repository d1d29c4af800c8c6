"""Randomised differential run of the TUPLE-cache attention forward (enable_duo_attention_eval's module forward,
reference llama.py:146-306) on the HIP backend against the oracle's ``tuple_forward_ref``: a bare attention module with
selector projections (q = hidden, k / v = column ranges of it — the construction of tests/golden/make_golden.py), random head
geometry, sink / recent, batch rows, prefill chunks and single-token steps.  After every call: attention output at the bar of
DESIGN §4, returned caches bit for bit (data movement + HF's bf16 rotary).

    python tests/fuzz_tuple_path.py --seconds 120 [--seed 1]"""
import argparse
import os
import random
import re
import sys
import time
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "duo-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import attn_close  # noqa: E402
from oracle.duo_oracle import tuple_forward_ref  # noqa: E402

D = 128
DEV = "cuda:0"


class Sel(torch.nn.Module):
    def __init__(self, lo, hi):
        super().__init__()
        self.lo, self.hi = lo, hi

    def forward(self, x):
        return x[..., self.lo:self.hi].clone()


def _hf_cos_sin(theta, pos0, S):
    inv_freq = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    freqs = torch.arange(pos0, pos0 + S)[None, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)


def _rms_only_on_a_small_sample(msg, n):
    """helpers.attn_close checks every element first and the statistical bar (rms err <= 2.5e-3 rms ref) last.  On a few
    hundred outputs the bf16 output rounding alone (0.6 - 2.3e-3 of an element, depending on where it lies in its binade)
    can put the SAMPLE rms a few per cent past 2.5e-3 — seen twice in 666 single-token cases, at 2.55 and 2.58e-3 on 512
    and 2048 outputs; that is sampling, not the kernel: accepted up to 2.5e-3 (1 + 4 / sqrt(2 n)) when every element passed."""
    m = re.match(r".*rms err ([0-9.e+-]+) vs rms\(ref\) ([0-9.e+-]+)", msg)
    return bool(m) and "out of tolerance" not in msg and n <= 8192 and \
        float(m.group(1)) <= 2.5e-3 * (1 + 4 / (2 * n) ** 0.5) * float(m.group(2))


def draw_case(rng):
    Hkv = rng.choice([1, 2, 4, 8])
    group = rng.choice([1, 2, 4, 4, 7] if Hkv < 8 else [1, 2, 4])
    chunk = lambda: rng.randint(1, 30) if rng.random() < 0.3 else rng.randint(31, 900)
    return dict(Hkv=Hkv, group=group, nf=rng.choice([0, Hkv, rng.randint(0, Hkv)]), sink=rng.choice([1, 4, 16, 64]),
                recent=rng.choice([2, 8, 32, 100, 256]), steps=[chunk() for _ in range(rng.randint(1, 3))] + [1] * rng.randint(0, 5),
                theta=rng.choice([1e4, 5e5, 1e6]), scale=rng.choice([0.5, 1.0, 1.0, 2.0]), B=rng.choice([1, 1, 2]),
                seed=rng.randint(0, 2 ** 31 - 1))


def run_case(c):
    from duo_attn.patch._duo import duo_attention_forward_one_way_reordered as fwd
    from duo_attn.patch._duo import release_tuple_arena
    from duo_attn.patch.tuple_kv_cache import hf_apply_rotary_pos_emb

    Hkv, group, nf, B = c["Hkv"], c["group"], c["nf"], c["B"]
    Hq = Hkv * group
    m = torch.nn.Module()
    m.config = types.SimpleNamespace(num_attention_heads=Hq, num_key_value_heads=Hkv, hidden_size=Hq * D)
    m.head_dim = D
    m.q_proj, m.k_proj, m.v_proj, m.o_proj = (torch.nn.Identity(), Sel(0, Hkv * D), Sel(Hq * D - Hkv * D, Hq * D),
                                              torch.nn.Identity())
    m.sink_size, m.recent_size = c["sink"], c["recent"]
    m.register_buffer("full_attention_heads", torch.tensor([1.0] * nf + [0.0] * (Hkv - nf)))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(c["seed"])
    past, ref_past, pos = None, None, 0
    try:
        for si, S in enumerate(c["steps"]):
            h = (torch.randn(B, S, Hq * D, generator=g) * c["scale"]).to(torch.bfloat16)
            cos, sin = _hf_cos_sin(c["theta"], pos, S)
            rep = (lambda t: t.repeat(B, 1, 1)) if (B > 1 and (c["seed"] + si) % 2) else (lambda t: t)     # HF hands [B, S, D]
            out, _, past = fwd(m, h.to(DEV), past_key_value=past, use_cache=True,
                               position_embeddings=(rep(cos).to(DEV), rep(sin).to(DEV)))
            q = h.clone().view(B, S, Hq, D)
            k = h[..., : Hkv * D].clone().view(B, S, Hkv, D)
            v = h[..., Hq * D - Hkv * D:].clone().view(B, S, Hkv, D)
            q, k = hf_apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=2)
            exact, ref_past, bud = tuple_forward_ref(q, k, v, ref_past, nf, c["sink"], c["recent"], round_p=False,
                                                     out_dtype=torch.float32, return_budget=True)
            what = f"step {si} (S={S}) pos {pos}"
            try:
                attn_close(out.view(B, S, Hq, D), exact, "", bud if S > 1 else None)
            except AssertionError as e:
                if not _rms_only_on_a_small_sample(str(e), out.numel()):
                    raise AssertionError(f"{what}: attention {e}") from None
            assert torch.equal(past[0].cpu(), ref_past[0]), what + ": retrieval cache"
            assert torch.equal(past[1].cpu(), ref_past[1]), what + ": streaming cache"
            pos += S
    finally:
        release_tuple_arena(m)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-oracle", action="store_true", help="harness / host-logic self-check: the oracle as device backend, on the CPU")
    a = ap.parse_args()
    if a.cpu_oracle:
        global DEV
        from duo_attn import backend
        from oracle.duo_oracle import OracleBackend

        DEV = "cpu"
        backend._set_backend_for_testing(OracleBackend(round_p=False))
    rng = random.Random(a.seed)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        c = draw_case(rng)
        n += 1
        try:
            run_case(c)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("FAIL", c, "\n    ", f"{type(e).__name__}: {str(e)[:500]}", flush=True)
            if not isinstance(e, AssertionError):
                import traceback

                traceback.print_exc()
    print(f"{n} cases in {time.time() - t0:.0f} s, {bad} failed (seed {a.seed})")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
