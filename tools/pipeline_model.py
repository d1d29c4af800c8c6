#!/usr/bin/env python3
"""What `bench.py --gpus N` should measure on an N-GPU node — a MODEL, not a measurement (no multi-GPU node has been
available in five rounds; the driver's SCALE run is the measurement).

The layer pipeline of bench.py (duo_attn/pipeline.py) replayed on paper from ONE-GPU measurements of the same build:

  * prefill: the job's row blocks (4096 rows on up to 4 GPUs, 2048 on more: bench.py's automatic choice) flow through the cost-balanced contiguous stages
    (``balanced_layer_split`` over the per-layer algorithmic FLOPs, as bench.py does); a block costs its algorithmic FLOPs on
    the stage's layers / (the measured one-GPU whole-job prefill rate x the measured efficiency of that block size), a
    hand-off costs block bytes / one xGMI link + a fixed latency; finish[s][b] = max(finish[s][b-1], finish[s-1][b] + hop)
    + cost[s][b];
  * decode (batch 1, autoregressive): the stages run one after the other — sum over layers of (fixed + bytes / stream
    rate), the two constants fitted to the measured duo / full-attention decode steps, plus a hop per stage boundary and the
    token's hop back.

    python tools/pipeline_model.py [--link-GBps 153] [--hop-us 15]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "duo-attention_amd")]

D, HQ, HKV, HIDDEN, W = 128, 32, 8, 4096, 384
G = HQ // HKV


def block_flops(nf, s, r0, n):
    """algorithmic attention FLOPs of rows [r0, r0 + n) of a chunk that starts at position s, one layer"""
    tri = n * (n + 1) / 2
    if s == 0:
        return G * HKV * 4 * D * (n * r0 + tri)
    full = 4 * D * (n * (s + r0) + tri)
    stream = 4 * D * (n * (min(s, W) + r0) + tri)
    return G * (nf * full + (HKV - nf) * stream)


def model_two_blocks(counts, ctx, chunk, world, rate, eff, link, hop, dec_fixed, dec_bw, n_decode):
    """the same job with TWO layer blocks per rank (duo_attn.pipeline.InterleavedLayerPipeline, bench.py --virtual-stages 2):
    rank r owns blocks r and world + r of ``interleaved_layer_split``; every rank works through (item, pass) units in groups
    of `world` items — pass 0 of a group, then its pass 1 — a unit starting when the rank is free AND its input has arrived"""
    from duo_attn.pipeline import interleaved_layer_split

    import bench

    pf = bench.prefill_flops(counts, ctx, chunk)
    layer_cost = [sum(row[l] for row in pf) for l in range(len(counts))]
    bounds, gs = interleaved_layer_split(layer_cost, world)
    rb = 4096 if world <= 4 else 2048
    blocks = [(s, r0, min(rb, min(chunk, ctx - s) - r0)) for s in range(0, ctx, chunk) for r0 in range(0, min(chunk, ctx - s), rb)]
    r = rate * eff[rb]
    n = len(blocks)
    seq = []
    for g0 in range(0, n, gs):
        items = range(g0, min(n, g0 + gs))
        seq += [(i, 0) for i in items] + [(i, 1) for i in items]
    cost = lambda rk, i, ps: sum(block_flops(counts[l], *blocks[i]) for l in range(*bounds[ps * world + rk])) / r
    done = {}           # (virtual stage, item) -> finish time
    free = [0.0] * world
    pos = [0] * world
    progressed = True
    while progressed:
        progressed = False
        for rk in range(world):
            while pos[rk] < len(seq):
                i, ps = seq[pos[rk]]
                vs = ps * world + rk
                if vs > 0 and (vs - 1, i) not in done:
                    break
                ready = done[(vs - 1, i)] + (blocks[i][2] * HIDDEN * 2 / link + hop) if vs > 0 else 0.0
                done[(vs, i)] = free[rk] = max(free[rk], ready) + cost(rk, i, ps)
                pos[rk] += 1
                progressed = True
    t_pre = done[(2 * world - 1, n - 1)]
    t_tok = sum(dec_fixed + x / dec_bw for x in bench.decode_bytes(counts, ctx)) + (2 * world - 1) * hop + hop
    busy = [sum(cost(rk, i, ps) for i in range(n) for ps in (0, 1)) / t_pre for rk in range(world)]
    return {"n_gpus": world, "row_block": rb, "virtual_stages": 2, "group_size": gs, "stages": bounds, "prefill_s": t_pre, "prefill_tok_s": ctx / t_pre,
            "decode_ms_per_token": t_tok * 1e3, "job_tok_s": (ctx + n_decode) / (t_pre + n_decode * t_tok),
            "stage_busy_min_max": [min(busy), max(busy)]}


def model(counts, ctx, chunk, world, rate, eff, link, hop, dec_fixed, dec_bw, n_decode):
    from duo_attn.utils import balanced_layer_split

    import bench

    pf = bench.prefill_flops(counts, ctx, chunk)
    layer_cost = [sum(row[l] for row in pf) for l in range(len(counts))]
    bounds = balanced_layer_split(layer_cost, world) if world > 1 else [(0, len(counts))]
    rb = chunk if world == 1 else (4096 if world <= 4 else 2048)
    blocks = [(s, r0, min(rb, min(chunk, ctx - s) - r0)) for s in range(0, ctx, chunk) for r0 in range(0, min(chunk, ctx - s), rb)]
    r = rate * eff[rb]
    finish = [[0.0] * len(blocks) for _ in range(world)]
    for st, (l0, l1) in enumerate(bounds):
        for b, (s, r0, n) in enumerate(blocks):
            cost = sum(block_flops(counts[l], s, r0, n) for l in range(l0, l1)) / r
            ready = finish[st - 1][b] + (n * HIDDEN * 2 / link + hop) if st else 0.0
            finish[st][b] = max(finish[st][b - 1] if b else 0.0, ready) + cost
    t_pre = finish[-1][-1]
    t_tok = sum(dec_fixed + x / dec_bw for x in bench.decode_bytes(counts, ctx)) + (world - 1) * hop + (hop if world > 1 else 0.0)
    busy = [sum(block_flops(counts[l], s, r0, n) for l in range(l0, l1) for s, r0, n in blocks) / r / t_pre for l0, l1 in bounds]
    return {"n_gpus": world, "row_block": rb, "stages": bounds, "prefill_s": t_pre, "prefill_tok_s": ctx / t_pre,
            "decode_ms_per_token": t_tok * 1e3, "job_tok_s": (ctx + n_decode) / (t_pre + n_decode * t_tok),
            "stage_busy_min_max": [min(busy), max(busy)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-GBps", type=float, default=153.0, help="one xGMI link (MI355X_MICROARCH.md)")
    ap.add_argument("--hop-us", type=float, default=15.0, help="fixed latency of one RCCL point-to-point hop (assumed)")
    ap.add_argument("--bench", default=os.path.join(ROOT, "profiles", "r6_a_bench.json"), help="the one-GPU bench line the rates come from")
    ap.add_argument("--eff", default="0.951,0.918", help="one-GPU efficiency of 4096- / 2048-row blocks against whole chunks (profiles/r6_prefill_plan.md)")
    a = ap.parse_args()
    import bench

    line = json.load(open(a.bench))
    counts, ctx, chunk, n_dec = bench.LLAMA3_8B_FULL_KV_HEADS, 131072, 16384, 128
    flops = sum(sum(r) for r in bench.prefill_flops(counts, ctx, chunk))
    rate = flops / (ctx / line["prefill_tok_s"])
    # decode: t = 32 fixed + bytes / bw, from the duo and the full-attention steps of the same line
    b_duo, b_full = sum(bench.decode_bytes(counts, ctx)), sum(bench.decode_bytes([HKV] * 32, ctx))
    t_duo, t_full = line["decode_ms_per_token"] * 1e-3, 1.0 / line["full_attention"]["decode_tok_s"]
    bw = (b_full - b_duo) / (t_full - t_duo)
    fixed = (t_duo - b_duo / bw) / 32
    e4, e2 = (float(x) for x in a.eff.split(","))
    eff = {chunk: 1.0, 4096: e4, 2048: e2}              # same-box one-GPU runs of this build (round 5: 0.935 / 0.872)
    out = {"inputs": {"one_gpu_prefill_PFLOPs": rate / 1e15, "decode_stream_TBps": bw / 1e12, "decode_fixed_us_per_layer": fixed * 1e6,
                      "block_efficiency": eff, "link_GBps": a.link_GBps, "hop_us": a.hop_us}, "rows": []}
    for world in (1, 2, 4, 8):
        out["rows"].append(model(counts, ctx, chunk, world, rate, eff, a.link_GBps * 1e9, a.hop_us * 1e-6, fixed, bw, n_dec))
    for world in (2, 4, 8):
        out["rows"].append(model_two_blocks(counts, ctx, chunk, world, rate, eff, a.link_GBps * 1e9, a.hop_us * 1e-6, fixed, bw, n_dec))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
