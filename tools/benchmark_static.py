#!/usr/bin/env python3
"""Model-level efficiency benchmark — the reference's eval/efficiency/benchmark_static.py protocol
(chunked prefill + single-token decode through the static dual KV cache, bench_func timing of
eval/efficiency/utils.py:7-30) on a RANDOM-INIT model of the named shape (no weights / tokenizer on
the box): input_ids = randint, mirroring the reference's content-free "a\\n\\n"*N prompt.

    python tools/benchmark_static.py --max_length 131072 --prefilling_chunk_size 16384 --sparsity 0.5
    python tools/benchmark_static.py --sparsity 0        # full attention through the same code

The whole HuggingFace model runs: embeddings, q/k/v/o and MLP GEMMs (hipBLASLt), this repo's HIP
RMSNorm / RoPE / pool updates / split-head attention, lm_head on the last position.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver stack (--pp / --tp)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SHAPES = {
    "llama-3-8b-1048k": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                             num_attention_heads=32, num_key_value_heads=8, vocab_size=128256,
                             max_position_embeddings=1048576, rope_theta=3580165449.0, rms_norm_eps=1e-5),
    "llama-2-7b-32k": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                           num_attention_heads=32, num_key_value_heads=32, vocab_size=32000,
                           max_position_embeddings=32768, rope_theta=10000.0, rms_norm_eps=1e-5,
                           rope_scaling={"rope_type": "linear", "factor": 8.0}),
    "mistral-7b-v0.2": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                            num_attention_heads=32, num_key_value_heads=8, vocab_size=32000,
                            max_position_embeddings=32768, rope_theta=1000000.0, rms_norm_eps=1e-5),
}


def bench_func(func, num_steps, num_warmup_steps):
    for _ in range(num_warmup_steps):
        func()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(num_steps):
        func()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / num_steps, torch.cuda.max_memory_allocated() / 1024 / 1024


def build_model(shape_name, device, seed):
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM

    from duo_attn.utils import seed_everything

    seed_everything(seed)
    shape = SHAPES[shape_name]
    is_mistral = shape_name.startswith("mistral")
    cfg_kw = dict(attn_implementation="eager", tie_word_embeddings=False, **shape)
    config = MistralConfig(sliding_window=None, **cfg_kw) if is_mistral else LlamaConfig(**cfg_kw)
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(device):
        model = (MistralForCausalLM if is_mistral else LlamaForCausalLM)(config)
    torch.set_default_dtype(torch.float32)
    return model.eval(), config, is_mistral


def run(args, quiet=False):
    """single GPU: the reference's benchmark_static protocol; returns the result dict"""
    from duo_attn.utils import sparsify_attention_heads

    say = (lambda *a: None) if quiet else print
    t0 = time.time()
    model, config, is_mistral = build_model(args.shape, "cuda", args.seed)
    say(f"random-init {args.shape}: {sum(p.numel() for p in model.parameters()) / 1e9:.2f} B params in {time.time() - t0:.1f} s")

    L, Hkv = config.num_hidden_layers, config.num_key_value_heads
    # no pattern files on the box: a synthetic importance matrix with the shipped patterns' structure
    # (sparsify_attention_heads then takes the global quantile exactly as the reference does)
    heads = np.random.RandomState(0).rand(L, Hkv)
    heads, sparsity = sparsify_attention_heads(heads, None, args.sparsity)
    if getattr(args, "pattern", None):
        # per-layer retrieval-head counts of a shipped pattern (bench.PATTERNS: recorded from the reference's TSVs)
        import bench

        counts = bench.PATTERNS[args.pattern][0]
        assert len(counts) == L and max(counts) <= Hkv
        heads = np.array([[1.0] * c + [0.0] * (Hkv - c) for c in counts])
        sparsity = 1.0 - sum(counts) / (L * Hkv)
    say(f"True Sparsity: {sparsity}")
    mod = __import__("duo_attn.patch." + ("mistral" if is_mistral else "llama"), fromlist=["x"])
    enable = getattr(mod, f"enable_{'mistral' if is_mistral else 'llama'}_duo_attention_static_kv_cache_eval")
    enable(model, heads)

    input_ids = torch.randint(0, config.vocab_size, (1, args.max_length - 1), device="cuda")
    max_size = input_ids.size(1) + 5
    C = args.prefilling_chunk_size
    kv_cache = mod.DuoAttentionStaticKVCache(model, heads, 1, max_size, args.sink_size, args.recent_size)

    def prefill():
        with torch.no_grad():
            for i in range(0, input_ids.size(1), C):
                out = model(input_ids=input_ids[:, i:i + C], past_key_values=kv_cache, use_cache=True)
        return out

    def func1():
        prefill()
        kv_cache.clear()

    ctx_latency, ctx_memory = bench_func(func1, args.prefill_steps, args.prefill_warmup)
    kv_cache.clear()
    out = prefill()
    pred = out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)

    from duo_attn import graph as duo_graph

    def eager_step():
        with torch.no_grad():
            model(input_ids=pred, past_key_values=kv_cache, use_cache=True)
        kv_cache.evict_last(1)

    def decode_fn(mode):
        """"eager": the reference's loop, every step issued from Python (the default); "loop": the SAME loop with
        DUO_AUTO_DECODE_GRAPH=1 — after two eager steps the step is captured and replayed behind the unchanged call
        (duo_attn.graph.auto_decode_step); "graph": the explicit DecodeStepGraph (evict_last inside the graph)"""
        duo_graph.AUTO_DECODE_GRAPH = mode == "loop"
        kv_cache._auto_graph = None
        kv_cache._decode_graph = None
        if mode != "graph":
            return eager_step
        from duo_attn.graph import DecodeStepGraph

        for _ in range(3):      # eager steps first: every kernel and GEMM handle is loaded before the capture
            eager_step()

        def step():
            with torch.no_grad():
                return model(input_ids=pred, past_key_values=kv_cache, use_cache=True, _duo_no_auto_graph=True).logits

        return DecodeStepGraph(kv_cache, step, evict_after=1).replay    # same protocol: one token, then evict_last(1), inside the graph

    auto_default = duo_graph.AUTO_DECODE_GRAPH
    main_mode = "graph" if args.graph else ("loop" if auto_default else "eager")
    gen_latency, gen_memory = bench_func(decode_fn(main_mode), args.decode_steps, args.decode_warmup)
    extra = {}
    for mode in ("eager", "loop"):
        if getattr(args, "all_decode_modes", False) and mode != main_mode:
            extra[mode], _ = bench_func(decode_fn(mode), args.decode_steps, args.decode_warmup)
    unfused_latency = None
    if getattr(args, "also_module_by_module", False):
        # the same decode step with the decoder layer run module by module (library GEMMs at M = 1, separate norm /
        # activation / add kernels) instead of the fused token-row linears (duo_attn/patch/_duo.py: duo_decode_layer_fused)
        from duo_attn.patch import _duo

        old = _duo._FUSED_DECODE_LAYER
        _duo._FUSED_DECODE_LAYER = False
        try:
            unfused_latency, _ = bench_func(decode_fn(main_mode), args.decode_steps, args.decode_warmup)
        finally:
            _duo._FUSED_DECODE_LAYER = old
    duo_graph.AUTO_DECODE_GRAPH = auto_default
    res = {
        "shape": args.shape, "context_length": args.max_length, "sparsity": float(sparsity),
        "prefilling_chunk_size": C, "avg_context_time_ms": ctx_latency, "prefill_tok_s": input_ids.size(1) / ctx_latency * 1e3,
        "avg_generation_time_ms": gen_latency, "decode_tok_s": 1e3 / gen_latency,
        "peak_context_memory_MB": ctx_memory, "peak_generation_memory_MB": gen_memory,
        "kv_cache_memory_MB": kv_cache.memory_usage / 1024 / 1024,
        "decode_mode": {"graph": "hip graph replay (explicit DecodeStepGraph)", "eager": "eager (the default: every step issued from Python)",
                        "loop": "the reference's unchanged loop, DUO_AUTO_DECODE_GRAPH=1 (auto-captured HIP graph behind model(...))"}[main_mode],
    }
    if unfused_latency is not None:
        res["avg_generation_time_module_by_module_ms"] = unfused_latency
    if "eager" in extra:
        res["avg_generation_time_eager_ms"] = extra["eager"]
    if "loop" in extra:
        res["avg_generation_time_reference_loop_ms"] = extra["loop"]
    del model, kv_cache
    torch.cuda.empty_cache()
    if getattr(args, "also_tuple", False):
        res["tuple_path"] = run_tuple(args, heads, quiet=quiet)
    return res


def run_tuple(args, heads, quiet=False):
    """The same protocol through the TUPLE cache (`enable_duo_attention_eval`, the API of the reference's README quick-start,
    NIAH and LongBench harnesses): chunked prefill with past_key_values tuples handed back and forth, then greedy decode
    steps (the tuple API has no evict_last: the context grows by one token per step)."""
    from duo_attn.patch import enable_duo_attention_eval
    from duo_attn.patch._duo import release_tuple_arena

    model, config, _ = build_model(args.shape, "cuda", args.seed)
    enable_duo_attention_eval(model, np.array(heads, dtype=float), args.sink_size, args.recent_size)
    input_ids = torch.randint(0, config.vocab_size, (1, args.max_length - 1), device="cuda")
    C = args.prefilling_chunk_size

    def prefill():
        past = None
        with torch.no_grad():
            for i in range(0, input_ids.size(1), C):
                out = model(input_ids=input_ids[:, i:i + C], past_key_values=past, use_cache=True)
                past = out.past_key_values
        return out

    def func1():
        prefill()
        release_tuple_arena(model)

    ctx_latency, ctx_memory = bench_func(func1, 1, 1)
    out = prefill()
    state = {"past": out.past_key_values, "tok": out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)}
    del out

    def step():
        with torch.no_grad():
            o = model(input_ids=state["tok"], past_key_values=state["past"], use_cache=True)
        state["past"] = o.past_key_values

    gen_latency, gen_memory = bench_func(step, args.decode_steps, args.decode_warmup)
    res = {"avg_context_time_ms": ctx_latency, "prefill_tok_s": input_ids.size(1) / ctx_latency * 1e3,
           "avg_generation_time_ms": gen_latency, "decode_tok_s": 1e3 / gen_latency,
           "peak_context_memory_MB": ctx_memory, "peak_generation_memory_MB": gen_memory,
           "what": "enable_duo_attention_eval (tuple cache): decode steps in the fused form — token-row linears, one launch for "
                   "HF rotary + cache updates, split-KV decode over the arena"}
    del model, state
    torch.cuda.empty_cache()
    return res


def _dist_setup():
    """one process per GPU over RCCL — or, with DUO_BENCH_DEBUG_SHARED_GPU=1, the same code path rehearsed on a ONE-GPU box:
    every rank computes on cuda:0, gloo group, hand-off and control tensors through host memory (not a measurement mode)"""
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    shared = os.environ.get("DUO_BENCH_DEBUG_SHARED_GPU") == "1"
    from duo_attn import launch

    launch.check_visible_gpus(world)
    local = 0 if shared else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if shared:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = f"cuda:{local}"
    return dist, rank, world, dev, ("cpu" if shared else dev), ("cpu" if shared else None)


def run_pp(args):
    """--pp: layer pipeline, one process per GPU (launch with torch.distributed.run).  The model is sharded with
    duo_attn.pipeline.PipelinedCausalLM: chunked prefill streamed through the stages (row blocks with --row_block),
    greedy decode with the token fed back from the last stage.  BASELINE cfg4's entry point:
        python tools/benchmark_static.py --pp --gpus 8 --max_length 1048576 --prefilling_chunk_size 32000 --row_block 4096
    (starts its own ranks; `python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/benchmark_static.py
    --pp ...` works as well)"""
    from duo_attn.pipeline import PipelinedCausalLM
    from duo_attn.utils import sparsify_attention_heads

    dist, rank, world, dev, ctl, handoff = _dist_setup()
    model, config, is_mistral = build_model(args.shape, dev, args.seed)     # same seed: same weights on every rank
    heads = np.random.RandomState(0).rand(config.num_hidden_layers, config.num_key_value_heads)
    heads, sparsity = sparsify_attention_heads(heads, None, args.sparsity)
    mod = __import__("duo_attn.patch." + ("mistral" if is_mistral else "llama"), fromlist=["x"])
    getattr(mod, f"enable_{'mistral' if is_mistral else 'llama'}_duo_attention_static_kv_cache_eval")(model, heads)
    pl = PipelinedCausalLM(model, heads, dev, handoff=handoff)
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(args.seed)
    input_ids = torch.randint(0, config.vocab_size, (1, args.max_length - 1), generator=g)
    kv = pl.make_kv_cache(1, input_ids.size(1) + args.decode_steps + args.decode_warmup + 5, args.sink_size, args.recent_size)

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t) / steps * 1e3

    rb = args.row_block if args.row_block > 0 else None

    def func1():
        pl.prefill(input_ids, kv, args.prefilling_chunk_size, row_block=rb)
        kv.clear()

    ctx_latency = timed(func1, args.prefill_steps, args.prefill_warmup)
    logits = pl.prefill(input_ids, kv, args.prefilling_chunk_size, row_block=rb)
    logits = pl.pp.broadcast_from_last(logits, (1, 1, config.vocab_size), torch.bfloat16)
    tok = logits[:, -1, :].argmax(-1, keepdim=True)
    pl.decode(tok, kv, args.decode_warmup)
    n = args.decode_steps
    gen_latency = timed(lambda: pl.decode(tok, kv, n), 1, 0) / n
    stage = torch.tensor([pl.pp.first_layer, pl.pp.last_layer, kv.memory_usage // (1 << 20),
                          int(torch.cuda.max_memory_allocated() // (1 << 20))], device=ctl, dtype=torch.int64)
    allst = [torch.zeros_like(stage) for _ in range(world)]
    dist.all_gather(allst, stage)
    res = {
        "mode": f"layer pipeline, {world} ranks ({dist.get_backend()} p2p" + (", shared-GPU rehearsal" if handoff else "") + ")", "shape": args.shape, "context_length": args.max_length,
        "sparsity": float(sparsity), "prefilling_chunk_size": args.prefilling_chunk_size, "row_block": rb,
        "avg_context_time_ms": ctx_latency, "prefill_tok_s": input_ids.size(1) / ctx_latency * 1e3,
        "avg_generation_time_ms": gen_latency, "decode_tok_s": 1e3 / gen_latency,
        "handoff_bytes_per_chunk": args.prefilling_chunk_size * config.hidden_size * 2,
        "stages": [{"rank": r, "layers": [int(s[0]), int(s[1])], "kv_cache_MB": int(s[2]), "peak_MB": int(s[3])}
                   for r, s in enumerate(allst)],
    }
    if rank == 0:
        print(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


def run_tp(args):
    """--tp: head-parallel tensor parallelism, one process per GPU (duo_attn.tp: retrieval heads dealt evenly over the
    ranks, column/row-sliced projections, two RCCL all-reduces of [1, S, hidden] per layer).  Same protocol as the
    single-GPU run; every rank executes every layer on its Hkv / tp heads."""
    from duo_attn.tp import shard_model_for_tp
    from duo_attn.utils import sparsify_attention_heads

    dist, rank, world, dev, ctl, _ = _dist_setup()
    model, config, is_mistral = build_model(args.shape, dev, args.seed)
    L, Hkv, hidden = config.num_hidden_layers, config.num_key_value_heads, config.hidden_size
    heads = np.random.RandomState(0).rand(L, Hkv)
    heads, sparsity = sparsify_attention_heads(heads, None, args.sparsity)
    mine = shard_model_for_tp(model, heads)
    torch.cuda.empty_cache()
    mod = __import__("duo_attn.patch." + ("mistral" if is_mistral else "llama"), fromlist=["x"])
    getattr(mod, f"enable_{'mistral' if is_mistral else 'llama'}_duo_attention_static_kv_cache_eval")(model, mine)
    input_ids = torch.randint(0, config.vocab_size, (1, args.max_length - 1), device=dev,
                              generator=torch.Generator(device=dev).manual_seed(args.seed))
    kv = mod.DuoAttentionStaticKVCache(model, mine, 1, input_ids.size(1) + 5, args.sink_size, args.recent_size)
    C = args.prefilling_chunk_size

    def prefill():
        with torch.no_grad():
            for i in range(0, input_ids.size(1), C):
                out = model(input_ids=input_ids[:, i:i + C], past_key_values=kv, use_cache=True)
        return out

    def func1():
        prefill()
        kv.clear()

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t) / steps * 1e3

    ctx_latency = timed(func1, args.prefill_steps, args.prefill_warmup)
    out = prefill()
    pred = out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)

    def func2():
        with torch.no_grad():
            model(input_ids=pred, past_key_values=kv, use_cache=True)
        kv.evict_last(1)

    gen_latency = timed(func2, args.decode_steps, args.decode_warmup)
    st = torch.tensor([int(mine.sum()), kv.memory_usage // (1 << 20)], device=ctl, dtype=torch.int64)
    allst = [torch.zeros_like(st) for _ in range(world)]
    dist.all_gather(allst, st)
    res = {
        "mode": f"head-parallel TP, {world} ranks ({dist.get_backend()} all-reduce" + (", shared-GPU rehearsal" if ctl == "cpu" else "") + ")", "shape": args.shape, "context_length": args.max_length,
        "sparsity": float(sparsity), "prefilling_chunk_size": C, "avg_context_time_ms": ctx_latency,
        "prefill_tok_s": input_ids.size(1) / ctx_latency * 1e3, "avg_generation_time_ms": gen_latency,
        "decode_tok_s": 1e3 / gen_latency, "all_reduces_per_layer": 2, "all_reduce_bytes_decode": hidden * 2,
        "all_reduce_bytes_prefill_chunk": C * hidden * 2,
        "ranks": [{"rank": r, "retrieval_kv_heads": int(s[0]), "kv_cache_MB": int(s[1])} for r, s in enumerate(allst)],
    }
    if rank == 0:
        print(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="llama-3-8b-1048k", choices=list(SHAPES))
    ap.add_argument("--max_length", type=int, default=131072)
    ap.add_argument("--prefilling_chunk_size", type=int, default=16384)
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--pattern", default=None, help="per-layer retrieval-head counts of a shipped pattern (a key of "
                    "bench.PATTERNS, e.g. mistral-7b-v0.2@raw) instead of a synthetic matrix at --sparsity")
    ap.add_argument("--sink_size", type=int, default=128)
    ap.add_argument("--recent_size", type=int, default=256)
    ap.add_argument("--prefill_steps", type=int, default=2)
    ap.add_argument("--prefill_warmup", type=int, default=1)
    ap.add_argument("--decode_steps", type=int, default=100)
    ap.add_argument("--decode_warmup", type=int, default=20)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--graph", action="store_true",
                    help="decode through duo_attn.graph.DecodeStepGraph (one captured step incl. evict_last, replayed)")
    ap.add_argument("--all_decode_modes", action="store_true",
                    help="also time the decode loop eagerly (the default) and with DUO_AUTO_DECODE_GRAPH=1 as the reference's unchanged loop "
                         "(auto-captured graph), next to the mode selected")
    ap.add_argument("--also_tuple", action="store_true",
                    help="also run the protocol through the tuple cache (enable_duo_attention_eval)")
    ap.add_argument("--also_module_by_module", action="store_true",
                    help="also time the decode step with the decoder layers run module by module (A/B of the fused layer form)")
    ap.add_argument("--pp", action="store_true", help="layer pipeline over the ranks of torch.distributed.run")
    ap.add_argument("--row_block", type=int, default=0, help="--pp: hand prefill chunks through the stages in row blocks")
    ap.add_argument("--tp", action="store_true", help="head-parallel tensor parallelism over the ranks of torch.distributed.run")
    ap.add_argument("--gpus", type=int, default=0,
                    help="--pp / --tp without a launcher: start this many ranks (one per GPU) under torch.distributed.run; "
                         "0 = every visible GPU.  Ignored when torch.distributed.run already started the ranks")
    return ap.parse_args(argv)


def main():
    args = parse()
    if args.pp or args.tp:
        from duo_attn import launch

        if not launch.launched_by_torchrun():
            # plain `python tools/benchmark_static.py --pp --gpus 8`: start the ranks here (one per GPU, RCCL)
            n = args.gpus or torch.cuda.device_count()
            if n < 2:
                raise SystemExit("--pp / --tp need at least two ranks: pass --gpus N (N GPUs visible, or "
                                 f"{launch.SHARED_GPU_ENV}=1 for the one-GPU rehearsal)")
            raise SystemExit(launch.self_launch(__file__, sys.argv[1:], n))
    if args.pp:
        return run_pp(args)
    if args.tp:
        return run_tp(args)
    res = run(args)
    # same fields as the reference's benchmark_result.txt (benchmark_static.py:108-119)
    print(f"Average generation time: {res['avg_generation_time_ms']:.4f} ms")
    print(f"Peak generation memory usage: {res['peak_generation_memory_MB']:.4f} MB")
    print(f"Average context time: {res['avg_context_time_ms']:.4f} ms")
    print(f"Peak context memory usage: {res['peak_context_memory_MB']:.4f} MB")
    print(f"Context length: {args.max_length}")
    print(f"Sparsity: {res['sparsity']}")
    print(f"Prefilling chunk size: {res['prefilling_chunk_size']}")
    print(f"KV cache memory usage: {res['kv_cache_memory_MB']:.4f} MB")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
