"""Randomised differential run of ``DuoAttentionStaticINT4KVCache`` (the reference's INT4 demo cache, demo/int4_kv.py:261-492
driven as demo/w8a8kv4_llama.py:219-278 does): random head geometry / pattern, sink / recent, batch rows; a few prefill chunks
(put -> chunked-prefill attention over the dequantised pools -> compress) and decode steps (put -> fused INT4 decode attention
-> compress) against the oracle's attention over oracle-dequantised rows, counters after every call.

    python tests/fuzz_int4_cache.py --seconds 90 [--seed 1]"""
import argparse
import os
import random
import sys
import time
import traceback

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "duo-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import ShapeModel, attn_close, heads_from_counts  # noqa: E402
from oracle.duo_oracle import flash_attn_func_ref  # noqa: E402
from oracle.int4_oracle import dequantize_int4_ref, quantize_int4_ref  # noqa: E402

DEV = "cuda:0"


def draw_case(rng):
    Hkv = rng.choice([1, 2, 4, 8])
    group = rng.choice([1, 2, 4, 4, 7] if Hkv < 8 else [1, 2, 4])
    L = rng.choice([1, 2])
    chunk = rng.choice([64, 200, 300, 512])
    return dict(Hkv=Hkv, group=group, counts=[rng.choice([0, Hkv, rng.randint(0, Hkv)]) for _ in range(L)],
                sink=rng.choice([4, 16, 128]), recent=rng.choice([8, 48, 256]), chunk=chunk,
                chunks=[rng.choice([1, 2, rng.randint(1, chunk), rng.randint(1, chunk)]) for _ in range(rng.randint(1, 4))],
                decode_steps=rng.randint(0, 4),
                B=rng.choice([1, 1, 2]), scale=rng.choice([0.5, 1.0, 1.0]), seed=rng.randint(0, 2 ** 31 - 1))


def run_case(c):
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache

    Hkv, G, counts, B, sink, recent = c["Hkv"], c["group"], c["counts"], c["B"], c["sink"], c["recent"]
    Hq, W = Hkv * G, sink + recent
    total = sum(c["chunks"]) + c["decode_steps"] + 2
    model = ShapeModel(len(counts), Hq, Hkv, 128, device=DEV, dtype=torch.float16)
    cache = DuoAttentionStaticINT4KVCache(model, heads_from_counts(counts, Hkv), B, total, sink, recent, c["chunk"])
    g = torch.Generator().manual_seed(c["seed"])
    dq = lambda x: torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(x.float().numpy())).astype(np.float32))
    hist = [dict(fk=torch.zeros(B, 0, nf, 128), fv=torch.zeros(B, 0, nf, 128), sk=torch.zeros(B, 0, Hkv - nf, 128),
                 sv=torch.zeros(B, 0, Hkv - nf, 128)) for nf in counts]
    kw = dict(round_p=False, out_dtype=torch.float32, return_budget=True)
    steps = list(c["chunks"]) + [1] * c["decode_steps"]
    for si, S in enumerate(steps):
        for l, nf in enumerate(counts):
            mk = lambda h: (torch.randn(B, S, h, 128, generator=g) * c["scale"]).to(torch.float16)
            q, k, v = mk(Hq), mk(Hkv), mk(Hkv)
            past = cache.kv_seq_len_list[l]
            decode = S == 1 and past > 0
            cache.put(l, k.to(DEV), v.to(DEV), dequantize=False)
            out = (cache.decode_attention(l, q.to(DEV)) if decode else cache.prefill_attention(l, q.to(DEV), k.to(DEV), v.to(DEV)))
            h = hist[l]
            h["fk"], h["fv"] = torch.cat([h["fk"], dq(k[:, :, :nf])], 1), torch.cat([h["fv"], dq(v[:, :, :nf])], 1)
            h["sk"], h["sv"] = torch.cat([h["sk"], dq(k[:, :, nf:])], 1), torch.cat([h["sv"], dq(v[:, :, nf:])], 1)
            ref, bud = torch.empty(B, S, Hq, 128), torch.empty(B, S, Hq, 128)
            if past == 0:
                ref, bud = flash_attn_func_ref(q, k, v, **kw)           # the first chunk attends to the raw chunk
            else:
                if nf:
                    ref[:, :, :nf * G], bud[:, :, :nf * G] = flash_attn_func_ref(q[:, :, :nf * G], h["fk"], h["fv"], **kw)
                if Hkv - nf:
                    ref[:, :, nf * G:], bud[:, :, nf * G:] = flash_attn_func_ref(q[:, :, nf * G:], h["sk"], h["sv"], **kw)
            what = f"step {si} (S={S}, {'decode' if decode else 'prefill'}) layer {l} past {past}"
            o = out.float().cpu().view(B, S, Hq, 128)
            if decode:
                err = (o - ref).abs()
                tol = 1e-3 * ref.abs() + 2.0 ** -10 * ref.abs() + 2.0 ** -10 * bud + 1e-3 * ref.pow(2).mean().sqrt()
                assert torch.isfinite(o).all() and (err <= tol).all(), f"{what}: {int((err > tol).sum())} out of tolerance, worst err/tol {float((err / tol).max()):.2f}"
            else:
                try:
                    attn_close(o, ref, "", bud)
                except AssertionError as e:
                    raise AssertionError(f"{what}: {e}") from None
            cache.compress(l)
            if h["sk"].shape[1] > W:
                h["sk"] = torch.cat([h["sk"][:, :sink], h["sk"][:, -recent:]], 1)
                h["sv"] = torch.cat([h["sv"][:, :sink], h["sv"][:, -recent:]], 1)
            assert cache.kv_seq_len_list[l] == h["fk"].shape[1], what + ": full counter"
            if Hkv - nf:
                assert cache.streaming_kv_seq_len_list[l] == h["sk"].shape[1], what + ": streaming counter"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
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
                traceback.print_exc()
    print(f"{n} cases in {time.time() - t0:.0f} s, {bad} failed (seed {a.seed})")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
