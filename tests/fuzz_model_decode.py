"""Randomised differential run at MODEL level: tiny random-init HuggingFace Llama / Mistral models of random geometry
(q / kv head counts, MLP width, layer count, projection biases, batch rows, head pattern, sink / recent, prompt length)
through the drop-in APIs, decode steps in every form the package has —

    static path:  module by module  |  fused decode layer, eager  |  the reference's loop with the automatic HIP graph
    tuple path (enable_duo_attention_eval):  module by module  |  fused decode layer

— within a cache format all forms must agree: logits within 2e-2 relative L2 of that format's module-by-module run (the fused
forms change only the summation order of the projections), greedy tokens >= 90 % equal where the reference's choice is not a tie, graph == eager bit for bit (both planned from the
same length bucket); across the two formats a sanity bound only (different arithmetic by design).

    python tests/fuzz_model_decode.py --seconds 120 [--seed 1]"""
import argparse
import copy
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

DEV = "cuda:0"
VOCAB = 211


def draw_case(rng):
    Hkv = rng.choice([1, 2, 2, 4])
    group = rng.choice([1, 2, 4] if Hkv < 4 else [1, 2])
    L = rng.choice([1, 2, 3])
    heads = [[float(rng.random() < 0.5) for _ in range(Hkv)] for _ in range(L)]
    return dict(family=rng.choice(["llama", "mistral"]), Hkv=Hkv, group=group, inter=8 * rng.randint(8, 300), L=L, heads=heads,
                bias=rng.random() < 0.3, B=rng.choice([1, 1, 2, 3]), sink=rng.choice([4, 16, 64]), recent=rng.choice([8, 48, 256]),
                # (a third of the prompts end just below a 64 * 2^k-row boundary: the decode steps cross it and the captured
                #  step is re-captured for the next length bucket, duo_attn/graph.py)
                prompt=rng.choice([rng.randint(2, 60), rng.randint(61, 400), rng.randint(401, 900),
                                   rng.choice([64, 128, 256, 512]) - rng.randint(1, 5)]), steps=rng.randint(3, 8),
                evict=rng.random() < 0.5, seed=rng.randint(0, 2 ** 31 - 1))


def build(c):
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM

    torch.manual_seed(c["seed"])
    Hq = c["Hkv"] * c["group"]
    kw = dict(hidden_size=Hq * 128, intermediate_size=c["inter"], num_hidden_layers=c["L"], num_attention_heads=Hq,
              num_key_value_heads=c["Hkv"], head_dim=128, vocab_size=VOCAB, max_position_embeddings=8192, rope_theta=500000.0,
              attn_implementation="eager", tie_word_embeddings=False)
    if c["family"] == "llama":
        m = LlamaForCausalLM(LlamaConfig(attention_bias=c["bias"], mlp_bias=c["bias"], **kw))
    else:
        m = MistralForCausalLM(MistralConfig(sliding_window=None, **kw))
    return m.to(torch.bfloat16).eval().to(DEV)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-9)).item()


def run_case(c):
    from duo_attn import graph
    from duo_attn.patch import _duo, enable_duo_attention_eval

    mod = __import__(f"duo_attn.patch.{c['family']}", fromlist=["x"])
    enable_static = getattr(mod, f"enable_{c['family']}_duo_attention_static_kv_cache_eval")
    base = build(c)
    heads = np.array(c["heads"])
    B, n_pre, n_dec = c["B"], c["prompt"], c["steps"]
    ids = torch.randint(0, VOCAB, (B, n_pre + n_dec), generator=torch.Generator().manual_seed(c["seed"] ^ 5)).to(DEV)
    old_fused, old_auto = _duo._FUSED_DECODE_LAYER, graph.AUTO_DECODE_GRAPH

    def static(fused, auto):
        model = copy.deepcopy(base)
        enable_static(model, heads.copy())
        cache = mod.DuoAttentionStaticKVCache(model, heads, B, n_pre + n_dec + 4, c["sink"], c["recent"])
        _duo._FUSED_DECODE_LAYER, graph.AUTO_DECODE_GRAPH = fused, auto
        out = []
        with torch.no_grad():
            model(input_ids=ids[:, :n_pre], past_key_values=cache, use_cache=True)
            for t in range(n_pre, n_pre + n_dec):
                # teacher-forced tokens, so that every form sees the same inputs; with `evict` the benchmark protocol
                # (reference benchmark_static.py:96-105): the step's row is dropped again and the next token replaces it
                o = model(input_ids=ids[:, t:t + 1], past_key_values=cache, use_cache=True)
                out.append(o.logits.float().cpu())
                if c["evict"]:
                    cache.evict_last(1)
        return torch.cat(out, 1), getattr(cache, "_decode_graph", None) is not None

    def tuple_path(fused):
        model = copy.deepcopy(base)
        enable_duo_attention_eval(model, heads.copy(), c["sink"], c["recent"])
        _duo._FUSED_DECODE_LAYER = fused
        out = []
        with torch.no_grad():
            o = model(input_ids=ids[:, :n_pre], past_key_values=None, use_cache=True)
            past = o.past_key_values
            for t in range(n_pre, n_pre + n_dec):
                o = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
                out.append(o.logits.float().cpu())
                if not c["evict"]:
                    past = o.past_key_values          # (evict: the same past again = the step's row dropped)
        _duo.release_tuple_arena(model)
        return torch.cat(out, 1)

    try:
        ref, _ = static(False, False)
        assert torch.isfinite(ref).all()
        forms = {"static fused eager": static(True, False)[0]}
        auto_logits, captured = static(True, True)
        forms["static reference loop, automatic graph"] = auto_logits
        if B == 1 and n_dec > 3:
            assert captured, "the reference loop's decode call was not graph-captured"
        forms["tuple module by module"] = tuple_path(False)
        forms["tuple fused"] = tuple_path(True)
    finally:
        _duo._FUSED_DECODE_LAYER, graph.AUTO_DECODE_GRAPH = old_fused, old_auto
    def close(name, lg, to, to_name):
        assert torch.isfinite(lg).all(), f"{name}: non-finite logits"
        r = _rel(lg, to)
        assert r < 2e-2, f"{name}: logits rel L2 {r:.3e} from {to_name}"
        # Greedy tokens: a random-init model's logits are nearly flat, so a position whose two leading reference logits lie
        # closer than 4x the rms logit difference of the two runs is a tie that either run may break either way — those
        # positions are not counted (seed 6004 of the round-6 run drew 2 such ties in 10 tokens); the others must agree >= 90 %.
        lf, tf = lg.float(), to.float()
        rms = (lf - tf).pow(2).mean().sqrt().item()
        top2 = tf.topk(2, dim=-1).values
        decided = (top2[..., 0] - top2[..., 1]) > 4 * rms
        same = lf.argmax(-1) == tf.argmax(-1)
        n_decided = int(decided.sum())
        agree = (same & decided).sum().item() / max(n_decided, 1)
        # (>= 0.9 up to float32 rounding of the mean: 9 of 10 tokens is 0.89999998)
        assert agree >= 0.9 - 1e-6 or n_decided < 10, \
            f"{name}: greedy tokens agree with {to_name} on {agree:.2f} of {n_decided} decided positions (rms logit difference {rms:.2e})"

    close("static fused eager", forms["static fused eager"], ref, "the module-by-module static run")
    close("static reference loop, automatic graph", forms["static reference loop, automatic graph"], ref, "the module-by-module static run")
    close("tuple fused", forms["tuple fused"], forms["tuple module by module"], "the module-by-module tuple run")
    if not c["evict"]:
        # (with the benchmark's evict_last the static pool has already slid by the dropped row — reference
        #  static_kv_cache.py:169-173 only rewinds the counters — so the two cache formats see different windows by design)
        # The two cache formats are different ARITHMETIC by the reference's design — HF rotary in bf16 (three roundings) and HF's
        # two-rounding RMSNorm on the tuple path, flashinfer's fp32 rotary and one-rounding norm on the static one — and a
        # random-init model's logits are nearly flat, so this is a sanity bound only (measured over 1 719 drawn models: 94 %
        # within 2e-2, worst 6.2e-2); the semantic equivalence has its own tests with real bars.
        r = _rel(forms["tuple module by module"], ref)
        assert r < 0.15, f"tuple path vs static path: logits rel L2 {r:.3e}"
    # the library plans eager and captured launches from the same length bucket (duo_decode_plan_bucket): the replayed steps
    # are the eager steps' kernels on the eager steps' grids — bit-equal, also across a bucket boundary
    assert torch.equal(forms["static reference loop, automatic graph"], forms["static fused eager"]), \
        f"automatic graph vs eager: rel {_rel(forms['static reference loop, automatic graph'], forms['static fused eager']):.3e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", default=None, help="re-run ONE drawn case (the dict a FAIL line printed) and exit")
    a = ap.parse_args()
    if a.case:
        import ast
        run_case(ast.literal_eval(a.case))
        print("case passed")
        return
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
