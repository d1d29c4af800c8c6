#!/usr/bin/env python3
"""Where a decode step's wall time goes at model level (whole random-init Llama-3-8B-shape HF model, 128K context):

    static   the reference's benchmark_static loop, eager (model(...) then kv_cache.evict_last(1))
    tuple    the enable_duo_attention_eval loop (past_key_values tuples handed back and forth)

For each: ms per token with a final synchronize (wall), ms per token of the host loop alone (enqueue time: the step is
host-bound when this is >= the wall figure), and a cProfile of the host loop.  The context is not prefilled — the cache
counters are set to the context length over zero pools (timing does not depend on the values).

    python tools/debug/decode_host_profile.py [static|tuple|both] [ctx=131072] [steps=40]
"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import benchmark_static as bs  # noqa: E402


def timed_loop(step, steps, warm=8):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    return t_wall / steps * 1e3, t_host / steps * 1e3


def profile(step, steps, top=22):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top)
    return "\n".join(l for l in s.getvalue().splitlines() if l.strip())


def heads_of(counts, hkv):
    return np.array([[1.0] * c + [0.0] * (hkv - c) for c in counts])


def static(ctx, steps, label=""):
    model, config, _ = bs.build_model("llama-3-8b-1048k", "cuda", 42)
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    heads = heads_of(bench.LLAMA3_8B_FULL_KV_HEADS, 8)
    enable_llama_duo_attention_static_kv_cache_eval(model, heads)
    kv = DuoAttentionStaticKVCache(model, heads, 1, ctx + 8, 128, 256)
    for l in range(kv.num_layers):
        kv.kv_seq_len_list[l] = ctx
        kv.streaming_kv_seq_len_list[l] = 384
    tok = torch.zeros(1, 1, dtype=torch.long, device="cuda")

    def step():
        with torch.no_grad():
            model(input_ids=tok, past_key_values=kv, use_cache=True)
        kv.evict_last(1)

    from duo_attn import graph

    for auto in (False, True):
        graph.AUTO_DECODE_GRAPH = auto
        wall, host = timed_loop(step, steps)
        print(f"[static{label}, {'auto-captured graph' if auto else 'eager (DUO_AUTO_DECODE_GRAPH=0)'}] {wall:.3f} ms/token wall, "
              f"{host:.3f} ms/token host loop ({ctx} ctx, reference loop unchanged)")
        print(profile(step, 10))
    del model, kv
    torch.cuda.empty_cache()


def tuple_path(ctx, steps):
    model, config, _ = bs.build_model("llama-3-8b-1048k", "cuda", 42)
    from duo_attn.patch import enable_duo_attention_eval

    counts = bench.LLAMA3_8B_FULL_KV_HEADS
    enable_duo_attention_eval(model, heads_of(counts, 8), 128, 256)
    past = tuple((torch.zeros(2, nf, ctx, 128, device="cuda", dtype=torch.bfloat16),
                  torch.zeros(2, 8 - nf, 384, 128, device="cuda", dtype=torch.bfloat16)) for nf in counts)
    tok = torch.zeros(1, 1, dtype=torch.long, device="cuda")
    state = {"past": past}

    def step():
        with torch.no_grad():
            state["past"] = model(input_ids=tok, past_key_values=state["past"], use_cache=True).past_key_values

    wall, host = timed_loop(step, steps)
    print(f"[tuple] {wall:.3f} ms/token wall, {host:.3f} ms/token host loop ({ctx}+ ctx, enable_duo_attention_eval loop)")
    print(profile(step, 10))
    del model, state, past
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    if which in ("static", "both"):
        static(ctx, steps)
    if which in ("tuple", "both"):
        tuple_path(ctx, steps)
