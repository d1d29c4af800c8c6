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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="llama-3-8b-1048k", choices=list(SHAPES))
    ap.add_argument("--max_length", type=int, default=131072)
    ap.add_argument("--prefilling_chunk_size", type=int, default=16384)
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--sink_size", type=int, default=128)
    ap.add_argument("--recent_size", type=int, default=256)
    ap.add_argument("--prefill_steps", type=int, default=2)
    ap.add_argument("--prefill_warmup", type=int, default=1)
    ap.add_argument("--decode_steps", type=int, default=100)
    ap.add_argument("--decode_warmup", type=int, default=20)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--graph", action="store_true",
                    help="decode through duo_attn.graph.DecodeStepGraph (one captured step incl. evict_last, replayed)")
    args = ap.parse_args()

    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM

    from duo_attn.utils import seed_everything, sparsify_attention_heads

    seed_everything(args.seed)
    shape = SHAPES[args.shape]
    is_mistral = args.shape.startswith("mistral")
    cfg_kw = dict(attn_implementation="eager", tie_word_embeddings=False, **shape)
    config = MistralConfig(sliding_window=None, **cfg_kw) if is_mistral else LlamaConfig(**cfg_kw)
    t0 = time.time()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device("cuda"):
        model = (MistralForCausalLM if is_mistral else LlamaForCausalLM)(config)
    torch.set_default_dtype(torch.float32)
    model.eval()
    print(f"random-init {args.shape}: {sum(p.numel() for p in model.parameters()) / 1e9:.2f} B params in {time.time() - t0:.1f} s")

    L, Hkv = config.num_hidden_layers, config.num_key_value_heads
    # no pattern files on the box: a synthetic importance matrix with the shipped patterns' structure
    # (sparsify_attention_heads then takes the global quantile exactly as the reference does)
    heads = np.random.RandomState(0).rand(L, Hkv)
    heads, sparsity = sparsify_attention_heads(heads, None, args.sparsity)
    print(f"True Sparsity: {sparsity}")
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

    def func2():
        with torch.no_grad():
            model(input_ids=pred, past_key_values=kv_cache, use_cache=True)
        kv_cache.evict_last(1)

    if args.graph:
        from duo_attn.graph import DecodeStepGraph

        for _ in range(3):      # eager steps first: every kernel and GEMM handle is loaded before the capture
            func2()

        def step():
            with torch.no_grad():
                return model(input_ids=pred, past_key_values=kv_cache, use_cache=True).logits

        graph = DecodeStepGraph(kv_cache, step, evict_after=1)
        func2 = graph.replay    # noqa: F811 — same protocol: one token, then evict_last(1), inside the graph
    gen_latency, gen_memory = bench_func(func2, args.decode_steps, args.decode_warmup)
    res = {
        "shape": args.shape, "context_length": args.max_length, "sparsity": float(sparsity),
        "prefilling_chunk_size": C, "avg_context_time_ms": ctx_latency, "prefill_tok_s": input_ids.size(1) / ctx_latency * 1e3,
        "avg_generation_time_ms": gen_latency, "decode_tok_s": 1e3 / gen_latency,
        "peak_context_memory_MB": ctx_memory, "peak_generation_memory_MB": gen_memory,
        "kv_cache_memory_MB": kv_cache.memory_usage / 1024 / 1024,
        "decode_mode": "hip graph replay" if args.graph else "eager",
    }
    # same fields as the reference's benchmark_result.txt (benchmark_static.py:108-119)
    print(f"Average generation time: {gen_latency:.4f} ms")
    print(f"Peak generation memory usage: {gen_memory:.4f} MB")
    print(f"Average context time: {ctx_latency:.4f} ms")
    print(f"Peak context memory usage: {ctx_memory:.4f} MB")
    print(f"Context length: {args.max_length}")
    print(f"Sparsity: {sparsity}")
    print(f"Prefilling chunk size: {C}")
    print(f"KV cache memory usage: {res['kv_cache_memory_MB']:.4f} MB")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
