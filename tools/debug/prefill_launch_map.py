#!/usr/bin/env python
"""Map of prefill launch efficiency by shape, and calibration of the launch planner's cost model.

For every shape (rows S of the query block, retrieval kv heads nf of 8, cached rows `past`, chunk rows seen so far r1) it times
ONE launch (HIP events, median of --reps) as the planner plans it, as the round-1..5 policy planned it (forced: class 0 only,
k <= 8 from rounds(long * k) / k), and for a sweep of forced (k0, k1) — next to the planner's own estimate of each, so the model
can be checked and its constants refitted (--fit prints a least-squares t_tile / t_fix / t_merge / t_part).

    python tools/debug/prefill_launch_map.py --rows 2048 --sweep            # row blocks of the layer pipeline
    python tools/debug/prefill_launch_map.py --rows 16384 --nf 1 4 8        # whole chunks
Algorithmic FLOPs per launch = 4 D (S * lenA + causal part) per q head; `eff` = the launch's FLOP rate over the
rate of a whole-chunk launch of an 8-retrieval-head layer on this box (measured first)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

HQ, HKV, D, G, W = 32, 8, 128, 4, 384
DEV = "cuda:0"


def legacy_k(long_wgs, min_tiles):
    best, bc = 1, 1e30
    for k in range(1, min(8, min_tiles) + 1):
        cost = ((long_wgs * k + 255) // 256) / k + 0.03 * (k - 1)
        if cost < bc - 1e-9:
            bc, best = cost, k
    return best


def flops(S, nf, past, r1):
    qoff = r1 - S
    causal = sum(i + qoff + 1 for i in range(S))
    per_full = 4 * D * (S * past + causal)
    per_str = 4 * D * (S * min(past, W) + causal)
    return G * (nf * per_full + (HKV - nf) * per_str)


class Bench:
    def __init__(self, max_past, max_rows):
        from duo_attn.backend import HipBackend

        self.be = HipBackend()
        g = torch.Generator(device=DEV).manual_seed(0)
        r = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
        self.pool_k, self.pool_v = r(HKV, max_past + max_rows, D), r(HKV, max_past + max_rows, D)    # head-major retrieval pools
        self.str_k, self.str_v = r(HKV, W, D), r(HKV, W, D)
        self.q = r(max_rows, HQ, D)
        self.ck, self.cv = r(max_rows, HKV, D), r(max_rows, HKV, D)       # the chunk's own rows (token-major, as in the forward)
        self.out = torch.empty_like(self.q)

    def launch(self, S, nf, past, r1):
        ns = HKV - nf
        q, out = self.q[:S], self.out[:S]
        full = stream = None
        if nf:
            pk, pv = self.pool_k[:nf].permute(1, 0, 2), self.pool_v[:nf].permute(1, 0, 2)
            full = (nf, 0, (pk[:past], pv[:past]) if past else None, (pk[past:past + r1], pv[past:past + r1]))
        if ns:
            sk, sv = self.str_k[:ns].permute(1, 0, 2), self.str_v[:ns].permute(1, 0, 2)
            n = min(past, W)
            stream = (ns, nf * G, (sk[:n], sv[:n]) if n else None, (self.ck[:r1, nf:], self.cv[:r1, nf:]))
        self.be.attention(q, out, G, full, stream, D ** -0.5)

    def time(self, S, nf, past, r1, flags, reps):
        from duo_attn import _hip

        _hip.set_debug_flags(flags)
        try:
            self.launch(S, nf, past, r1)
            plan = _hip.last_prefill_plan()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.launch(S, nf, past, r1)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
        finally:
            _hip.set_debug_flags(0)
        ts.sort()
        return ts[len(ts) // 2], plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[2048])
    ap.add_argument("--nf", type=int, nargs="+", default=[1, 2, 3, 4, 5, 6, 8])
    ap.add_argument("--past", type=int, nargs="+", default=[16384, 114688])
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sweep", action="store_true", help="also a sweep of forced (k0, k1)")
    ap.add_argument("--fit", action="store_true")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    from duo_attn import _hip

    bench = Bench(max(a.past), a.chunk)
    # the box's whole-chunk rate: 8 retrieval heads, 16384 rows, long pool
    t_ref, _ = bench.time(a.chunk, 8, max(a.past), a.chunk, 0, a.reps)
    ref_rate = flops(a.chunk, 8, max(a.past), a.chunk) / t_ref
    print(f"# whole-chunk reference: {t_ref / 1e3:.3f} ms = {ref_rate / 1e6:.1f} TFLOP/s", flush=True)
    rows = []
    for S in a.rows:
        for past in a.past:
            for nf in a.nf:
                for r1 in sorted({S, a.chunk}):
                    if r1 < S:
                        continue
                    f = flops(S, nf, past, r1)
                    t_auto, plan = bench.time(S, nf, past, r1, 0, a.reps)
                    nq = (S + 255) // 256
                    min_tiles = (past + 63) // 64 + 1
                    kl = legacy_k(nf * G * nq, min_tiles) if nf else 1
                    t_leg, _ = bench.time(S, nf, past, r1, (kl << 12) | (1 << 16), a.reps)
                    rec = {"S": S, "nf": nf, "past": past, "r1": r1, "auto_us": t_auto, "plan": plan[:2], "est_us": plan[2],
                           "legacy_us": t_leg, "legacy_k": kl, "eff_auto": f / t_auto / ref_rate, "eff_legacy": f / t_leg / ref_rate}
                    if a.sweep:
                        sw = {}
                        for k0 in (1, 2, 4, 8, 16):
                            for k1 in (1, 2, 4):
                                if not nf and k0 > 1 or nf == HKV and k1 > 1:
                                    continue
                                t, p = bench.time(S, nf, past, r1, (k0 << 12) | (k1 << 16), a.reps)
                                est = _hip.prefill_plan(nf, HKV - nf, G, S, past, r1, min(past, W), r1, force=p[0] | (p[1] << 8))
                                sw[f"{p[0]},{p[1]}"] = (round(t, 1), round(est["est_us"] or 0, 1))
                        rec["sweep"] = sw
                    rows.append(rec)
                    print(json.dumps(rec), flush=True)
    tot_a, tot_l, tot_f = sum(r["auto_us"] for r in rows), sum(r["legacy_us"] for r in rows), sum(flops(r["S"], r["nf"], r["past"], r["r1"]) for r in rows)
    print(f"# all shapes: planner {tot_f / tot_a / ref_rate:.3f}, legacy {tot_f / tot_l / ref_rate:.3f} of the whole-chunk rate", flush=True)
    if a.json:
        with open(a.json, "w") as fjs:
            json.dump({"ref_tflops": ref_rate / 1e6, "rows": rows}, fjs)


if __name__ == "__main__":
    main()
