"""GPU-side cost of the 32-layer op-level decode step by context length: the two-launch step (scan + merge) against the
single-launch step (duo_decode_step_bf16), each as the eager loop bench.py times AND as a captured graph (DecodeStepGraph:
no host in the loop) — at <= 32K the eager loop is host-bound (its time does not depend on the context), so only the replayed
figure says what the kernels cost.
    python tools/debug/decode_graph_sweep.py [--pattern mistral|llama3] [--ctx 4096 16384 32768 65536 131072]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, n, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pattern", default="mistral-7b-v0.2@0.5")
    ap.add_argument("--ctx", type=int, nargs="+", default=[4096, 16384, 32768, 65536, 131072])
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    from duo_attn import _hip
    from duo_attn.graph import DecodeStepGraph

    dev = torch.device("cuda", 0)
    counts = bench.PATTERNS[a.pattern][0]
    for ctx in a.ctx:
        hp = bench.HotPath(counts, (0, len(counts)), ctx, min(ctx, 16384), dev)
        for l in range(len(counts)):          # pretend the context is cached (contents irrelevant for timing)
            hp.cache.kv_seq_len_list[l] = ctx
            hp.cache.streaming_kv_seq_len_list[l] = bench.SINK + bench.RECENT
        row = {"ctx": ctx}
        for name, two in (("two_launch", True), ("one_launch", False)):
            _hip.decode_layer.__defaults__ = (two,)
            _hip.decode_layer_dev.__defaults__ = (two,)
            row[name + "_eager_ms"] = timed(lambda: hp.decode_stage(0, None), a.steps)

            def step():
                for li in range(len(counts)):
                    hp.layer_core(li, 1, ctx, hp.q_1, hp.k_1, hp.v_1)

            g = DecodeStepGraph(hp.cache, step, evict_after=1)
            row[name + "_graph_ms"] = timed(g.replay, a.steps)
            del g
        by = sum(bench.decode_bytes(counts, ctx))
        if by:
            row["bytes"] = by
            row["graph_frac_two"] = by / (row["two_launch_graph_ms"] * 1e-3) / 8e12
            row["graph_frac_one"] = by / (row["one_launch_graph_ms"] * 1e-3) / 8e12
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
