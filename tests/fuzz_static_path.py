"""Randomised differential run of the static hot path: HIP kernels (through the C ABI) against the oracle.

Not collected by pytest (no ``test_`` prefix): ``tests/test_fuzz_gpu.py`` runs a fixed handful of seeds of it under ``-m gpu``;
as a script it keeps drawing cases for a time budget and lists every case that left the parity bar —

    python tests/fuzz_static_path.py --seconds 240 [--seed 1] [--big]

A case = random head geometry (any GQA group size, ragged per-layer retrieval counts incl. 0 and all), sink / recent,
RoPE base, data scale, batch rows, a few prefill chunks of random length (whole, or as row blocks of the layer pipeline)
and decode steps with or without the benchmark's ``evict_last(1)``.  Checked after every call: the attention output
(``helpers.attn_close``, the bar of DESIGN §4), the counters, V pools bit for bit, K pools to one bf16 ulp on < 1 % of the
elements (device sincos).  The reference lines restated by the oracle: llama.py:309-434, static_kv_cache.py:60-167."""
import argparse
import copy
import os
import random
import re
import sys
import time
import traceback

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "duo-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import ShapeModel, attn_close, heads_from_counts  # noqa: E402
from oracle.duo_oracle import StaticCacheRef, static_forward_ref  # noqa: E402

D = 128
DEV = "cuda:0"


def _ulp_close(ours, ref, what, max_frac=0.01):
    o, r = ours.cpu().float(), ref.cpu().float()
    assert o.shape == r.shape, (what, o.shape, r.shape)
    if o.numel() == 0:
        return
    diff = (o - r).abs()
    tol = torch.clamp(torch.maximum(o.abs(), r.abs()) * 2.0 ** -7, min=1e-5)
    assert (diff <= tol).all(), f"{what}: max diff {diff.max():.3e} exceeds one bf16 ulp"
    frac = (diff > 0).float().mean().item()
    assert frac <= max(max_frac, 4.0 / o.numel()), f"{what}: {frac:.4%} elements differ"


def _rms_only_on_a_small_sample(msg, n):
    """helpers.attn_close checks every element first and the statistical bar (rms err <= 2.5e-3 rms ref) last.  On a few
    hundred outputs the bf16 output rounding alone (0.6 - 2.3e-3 of an element, depending on where it lies in its binade)
    can put the SAMPLE rms a few per cent past 2.5e-3 — seen twice in 666 single-token cases, at 2.55 and 2.58e-3 on 512
    and 2048 outputs; that is sampling, not the kernel: accepted up to 2.5e-3 (1 + 4 / sqrt(2 n)) when every element passed."""
    m = re.match(r".*rms err ([0-9.e+-]+) vs rms\(ref\) ([0-9.e+-]+)", msg)
    return bool(m) and "out of tolerance" not in msg and n <= 8192 and \
        float(m.group(1)) <= 2.5e-3 * (1 + 4 / (2 * n) ** 0.5) * float(m.group(2))


def _no_noisier_than_the_reference_arithmetic(msg, out, exp, before, v, l, pos, c):
    """The statistical bar (rms err <= 2.5e-3 rms ref against EXACT-P attention) is the noise of the reference's arithmetic —
    P rounded to bf16 before P.V, bf16 output (FA2) — on N(0, 1) data: 2.33e-3.  That noise is data dependent: the relative
    error of one bf16 rounding is twice as large just above a power of two as just below it, and with low-variance scores
    (data scaled by 0.5: a flat softmax whose weights straddle 0.5) the reference's own arithmetic sits at 2.4 - 2.55e-3 (case
    483580883 of seed 9005, round 5: the HIP kernel 2.549e-3, the oracle's bf16-P form on the same inputs 2.550e-3).  So when
    every ELEMENT passed and only the rms bar is crossed on a prefill call, the call is held to the reference arithmetic
    itself: rms error no more than ``helpers.NOISE_RATIO_BAR`` times that of the oracle's bf16-P / bf16-output form on the same inputs."""
    if before is None or "out of tolerance" in msg or "rms err" not in msg:
        return False
    ref0, q0, k0 = before
    exp_p = static_forward_ref(q0, k0, v, ref0, l, pos, c["rope_scale"], c["theta"], round_p=True, out_dtype=torch.bfloat16)
    if isinstance(exp_p, tuple):
        exp_p = exp_p[0]
    e = exp.float().cpu()
    ours, theirs = (out.float().cpu() - e).pow(2).mean().sqrt(), (exp_p.float().cpu() - e).pow(2).mean().sqrt()
    from helpers import NOISE_RATIO_BAR         # (measured over the fixed suite: median 1.03, max 1.09 — helpers.attn_close)

    return bool(ours <= NOISE_RATIO_BAR * theirs)


def draw_case(rng: random.Random, big=False):
    Hkv = rng.choice([1, 2, 3, 4, 8])
    group = rng.choice([1, 2, 3, 4, 4, 5, 6, 7, 8] if Hkv <= 4 else [1, 2, 4])
    L = rng.choice([1, 2])
    counts = [rng.choice([0, Hkv, rng.randint(0, Hkv), rng.randint(0, Hkv)]) for _ in range(L)]
    sink = rng.choice([1, 2, 4, 16, 64, 128])
    recent = rng.choice([1, 3, 8, 32, 100, 256, 300])

    def chunk_len():
        r = rng.random()
        if r < 0.25:
            return rng.randint(1, 40)
        if r < 0.85:
            return rng.randint(41, 700)
        return rng.randint(701, 5000 if big else 1800)

    chunks = [chunk_len() for _ in range(rng.randint(1, 4))]
    blocks = None
    if rng.random() < 0.3:
        blocks = rng.choice([17, 64, 128, 200, 256, 512])
    return dict(Hkv=Hkv, group=group, counts=counts, sink=sink, recent=recent, chunks=chunks, row_block=blocks,
                decode_steps=rng.randint(0, 4), evict=rng.random() < 0.5, graph=rng.random() < 0.35, theta=rng.choice([1e4, 5e5, 1e6, 3580165449.0]),
                rope_scale=rng.choice([1.0, 1.0, 4.0]), scale=rng.choice([0.5, 1.0, 1.0, 2.5]), B=rng.choice([1, 1, 1, 2]),
                seed=rng.randint(0, 2 ** 31 - 1))


def run_case(c):
    """raises AssertionError (with the failing call named) if the HIP path leaves the bar"""
    from duo_attn.patch._duo import duo_static_attention_core, duo_static_attention_row_block
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    Hkv, group, counts, B = c["Hkv"], c["group"], c["counts"], c["B"]
    Hq, L = Hkv * group, len(counts)
    heads = heads_from_counts(counts, Hkv)
    total = sum(c["chunks"]) + c["decode_steps"] + 3
    cache = DuoAttentionStaticKVCache(ShapeModel(L, Hq, Hkv, D, device=DEV), heads, B, total, c["sink"], c["recent"])
    ref = StaticCacheRef(L, Hkv, D, heads, B, total, c["sink"], c["recent"])
    g = torch.Generator().manual_seed(c["seed"])
    mk = lambda S, h: (torch.randn((B, S, h, D), generator=g) * c["scale"]).to(torch.bfloat16)
    dev = lambda t: t.to(DEV, copy=True)        # (both sides rotate q and k in place)

    # The oracle rotates with the host's sincos, the device with its own: the bf16 results differ by one ulp on a fraction
    # of a per cent of the elements, and with large-magnitude data one flipped ulp of a key moves a score by ~0.01 — the
    # attention bar would then measure the RoPE rounding, not the attention.  As in the kernel tests, the two are checked
    # separately: device rotation against the host's to one ulp, and the oracle's attention on the device-rotated rows.
    import oracle.duo_oracle as O
    from duo_attn.patch.flashinfer_utils import apply_rope_inplace

    host_rope = O.apply_rope_inplace_ref

    def rope_checked(q, k, offsets, rope_scale, rope_theta):
        qh, kh = q.clone(), k.clone()
        host_rope(qh, kh, offsets, rope_scale, rope_theta)
        qd, kd = dev(q), dev(k)
        apply_rope_inplace(qd, kd, offsets, rope_scale, rope_theta)
        _ulp_close(qd, qh, "RoPE of q")
        _ulp_close(kd, kh, "RoPE of k")
        q.copy_(qd.cpu())
        k.copy_(kd.cpu())
        return q, k

    O.apply_rope_inplace_ref = rope_checked
    try:
        _run_steps(c, cache, ref, mk, dev, duo_static_attention_core, duo_static_attention_row_block)
    finally:
        O.apply_rope_inplace_ref = host_rope


def _run_steps(c, cache, ref, mk, dev, duo_static_attention_core, duo_static_attention_row_block):
    Hkv, L = c["Hkv"], len(c["counts"])
    Hq = Hkv * c["group"]
    pos = 0

    def check(l, out, q, k, v, S, what, counters=True):
        """the oracle's call for the same inputs, then outputs / counters / pools"""
        before = (copy.deepcopy(ref), q.clone(), k.clone()) if S > 1 else None     # (the oracle's call appends and rotates in place)
        exp, bud = static_forward_ref(q, k, v, ref, l, pos, c["rope_scale"], c["theta"], round_p=False,
                                      out_dtype=torch.float32, return_budget=True)
        try:
            attn_close(out, exp, "", bud if S > 1 else None)
        except AssertionError as e:
            if not (_rms_only_on_a_small_sample(str(e), out.numel()) or
                    _no_noisier_than_the_reference_arithmetic(str(e), out, exp, before, v, l, pos, c)):
                raise AssertionError(f"{what}: attention {e}") from None
        n, m = ref.kv_seq_len_list[l], ref.streaming_kv_seq_len_list[l]
        assert not counters or (cache.kv_seq_len_list[l] == n and cache.streaming_kv_seq_len_list[l] == m), what + ": counters"
        _ulp_close(cache.full_key_states_list[l][:, :n], ref.full_key_states_list[l][:, :n], what + ": full K pool")
        assert torch.equal(cache.full_value_states_list[l][:, :n].cpu(), ref.full_value_states_list[l][:, :n]), what + ": full V pool"
        _ulp_close(cache.streaming_key_states_list[l][:, :m], ref.streaming_key_states_list[l][:, :m], what + ": stream K pool")
        assert torch.equal(cache.streaming_value_states_list[l][:, :m].cpu(),
                           ref.streaming_value_states_list[l][:, :m]), what + ": stream V pool"
        ref.full_key_states_list[l][:, :n].copy_(cache.full_key_states_list[l][:, :n].cpu())
        ref.streaming_key_states_list[l][:, :m].copy_(cache.streaming_key_states_list[l][:, :m].cpu())

    steps = [(S, False) for S in c["chunks"]] + [(1, c["evict"])] * c["decode_steps"]
    graph = None
    for si, (S, evict) in enumerate(steps):
        decode = si >= len(c["chunks"])
        if graph is not None:
            # a replay of the captured step (device-side lengths, the eviction inside the graph): all layers at once
            data = [(mk(1, Hq), mk(1, Hkv), mk(1, Hkv)) for _ in range(L)]
            for l, (q, k, v) in enumerate(data):
                g_q[l].copy_(q); g_k[l].copy_(k); g_v[l].copy_(v)
            graph.replay()
            outs = [o.clone() for o in g_out]
            for l, (q, k, v) in enumerate(data):
                # (with the eviction inside the graph the host counters are already one step on: compared after the oracle's own)
                check(l, outs[l], q, k, v, 1, f"step {si} (graph replay) layer {l} pos {pos}", counters=not evict)
            if evict:
                ref.evict_last(1)
                assert cache.kv_seq_len_list == ref.kv_seq_len_list and \
                    cache.streaming_kv_seq_len_list == ref.streaming_kv_seq_len_list, f"step {si} (graph replay): counters after evict_last"
            else:
                pos += 1
            continue
        for l in range(L):
            q, k, v = mk(S, Hq), mk(S, Hkv), mk(S, Hkv)
            rb = c["row_block"] if (S > 1 and not decode) else None
            if rb and rb < S:
                parts = [duo_static_attention_row_block(dev(q[:, r0:r0 + rb]), dev(k[:, r0:r0 + rb]),
                                                        dev(v[:, r0:r0 + rb]), cache, l, r0, S, c["rope_scale"], c["theta"])
                         for r0 in range(0, S, rb)]
                out = torch.cat(parts, 1)
            else:
                out = duo_static_attention_core(dev(q), dev(k), dev(v), cache, l, pos, c["rope_scale"], c["theta"])
            check(l, out, q, k, v, S, f"step {si} (S={S}{' row blocks of %d' % rb if rb and rb < S else ''}) layer {l} pos {pos}")
        if evict:
            cache.evict_last(1)
            ref.evict_last(1)
        else:
            pos += S
        if decode and c.get("graph") and graph is None and DEV != "cpu" and si + 1 < len(steps):
            # the first decode step ran eagerly (library handles, workspaces); the rest are replays of ONE captured step
            from duo_attn.graph import DecodeStepGraph

            B = c["B"]
            z = lambda h: torch.zeros(B, 1, h, D, dtype=torch.bfloat16, device=DEV)
            g_q, g_k, g_v, g_out = [z(Hq) for _ in range(L)], [z(Hkv) for _ in range(L)], [z(Hkv) for _ in range(L)], [z(Hq) for _ in range(L)]

            def step_fn():
                for l in range(L):
                    g_out[l].copy_(duo_static_attention_core(g_q[l], g_k[l], g_v[l], cache, l, None, c["rope_scale"], c["theta"]))
                return g_out

            graph = DecodeStepGraph(cache, step_fn, evict_after=1 if evict else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--cpu-oracle", action="store_true", help="harness self-check: the oracle as device backend, on the CPU")
    a = ap.parse_args()
    if a.cpu_oracle:
        global DEV
        from duo_attn import backend
        from oracle.duo_oracle import OracleBackend

        DEV = "cpu"
        backend._set_backend_for_testing(OracleBackend(round_p=False))
    rng = random.Random(a.seed)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < a.seconds:
        c = draw_case(rng, a.big)
        n += 1
        try:
            run_case(c)
        except Exception as e:      # noqa: BLE001 - every failure is reported with its case
            bad.append((c, f"{type(e).__name__}: {e}"))
            print("FAIL", c, "\n    ", f"{type(e).__name__}: {str(e)[:600]}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
    print(f"{n} cases in {time.time() - t0:.0f} s, {len(bad)} failed (seed {a.seed})")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
