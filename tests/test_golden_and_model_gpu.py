"""GPU: (1) the HIP hot path against the golden outputs of the reference's own forward
(tests/golden/static_*.npz); (2) the patch API on real HuggingFace Llama/Mistral modules on the GPU,
HIP backend end to end (attention, RoPE, RMSNorm, pool updates) against the unpatched HF model."""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import ShapeModel, attn_close, heads_from_counts
from oracle.duo_oracle import StaticCacheRef, static_forward_ref
from test_oracle_golden import bf16, load, split_hidden, ulp_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["static_a.npz", "static_b.npz", "static_c.npz"])
def test_hip_hot_path_reproduces_reference_golden(name):
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    g = load(name)
    Hq, Hkv, D, sink, recent = (int(x) for x in g["dims"])
    theta, factor = (float(x) for x in g["rope"])
    counts = [int(c) for c in g["counts"]]
    steps = [int(s) for s in g["steps"]]
    n_prefill = int(g["n_prefill"])
    total = sum(steps) + 2
    heads = heads_from_counts(counts, Hkv)
    # static_c: the reference's batch dimension (2 rows in ONE batched launch here), rows at different RoPE positions
    starts = [int(x) for x in g["starts"]] if "starts" in g.files else [0]
    B = len(starts)
    cache = DuoAttentionStaticKVCache(ShapeModel(len(counts), Hq, Hkv, D, device=DEV), heads, B, total, sink, recent)
    ref = StaticCacheRef(len(counts), Hkv, D, heads, B, total, sink, recent)
    pos = 0
    for si, S in enumerate(steps):
        for l in range(len(counts)):
            q, k, v = split_hidden(bf16(g[f"h_{si}_{l}"]), Hq, Hkv, D)
            p0 = pos if B == 1 else [pos + s0 for s0 in starts]
            out = duo_static_attention_core(q.to(DEV), k.to(DEV), v.to(DEV), cache, l, p0, factor, theta)
            exact, bud = static_forward_ref(q, k, v, ref, l, p0, factor, theta, round_p=False,
                                            out_dtype=torch.float32, return_budget=True)
            golden = bf16(g[f"o_{si}_{l}"]).view(B, S, Hq, D).float()
            # (1) against the exact-P fp32 oracle on the same inputs: the usual bar
            attn_close(out, exact, f"{name} step {si} layer {l}", bud if S > 1 else None)
            # (2) against the reference's own bf16 output: both sides carry one output rounding -> allow ONE more
            #     bf16 ulp of the golden value than (1), nothing else
            err = (out.float().cpu() - golden).abs()
            tol = (2.0 ** -8) * golden.abs() * 2 + 1e-3 * golden.abs() + (2.0 ** -8) * (bud if S > 1 else 0 * bud) \
                + 1e-3 * golden.pow(2).mean().sqrt()
            assert (err <= tol).all(), f"{name} step {si} layer {l}: {int((err > tol).sum())} elements beyond the golden bar"
        if si >= n_prefill:
            cache.evict_last(1)
            ref.evict_last(1)
        else:
            pos += S
    for l in range(len(counts)):
        n, m = (int(x) for x in g[f"len_{l}"])
        assert cache.kv_seq_len_list[l] == n and cache.streaming_kv_seq_len_list[l] == m
        assert torch.equal(cache.full_value_states_list[l][:, :n].cpu(), bf16(g[f"fullv_{l}"]))
        assert torch.equal(cache.streaming_value_states_list[l][:, :m].cpu(), bf16(g[f"strv_{l}"]))
        # RoPE'd keys: per element within ONE bf16 ulp of the reference's value, on a small fraction of elements
        # (fp64 angle in the generator's stub vs fp32 angle here, device vs host sincos last bit) — the same bar the
        # CPU twin holds the oracle to (test_oracle_golden.ulp_close), not an absolute bound
        ulp_close(cache.full_key_states_list[l][:, :n].cpu(), bf16(g[f"fullk_{l}"]), f"full K {l}", max_frac=0.05)
        ulp_close(cache.streaming_key_states_list[l][:, :m].cpu(), bf16(g[f"strk_{l}"]), f"stream K {l}", max_frac=0.05)


def tiny(family, seed=0):
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM

    torch.manual_seed(seed)
    kw = dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
              num_key_value_heads=2, head_dim=128, vocab_size=211, max_position_embeddings=8192,
              rope_theta=500000.0, attn_implementation="eager", tie_word_embeddings=False)
    if family == "llama":
        m = LlamaForCausalLM(LlamaConfig(**kw))
    else:
        m = MistralForCausalLM(MistralConfig(sliding_window=None, **kw))
    return m.to(torch.bfloat16).eval().to(DEV)


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_static_path_on_gpu_matches_hf(family):
    mod = __import__(f"duo_attn.patch.{family}", fromlist=["x"])
    enable_static = getattr(mod, f"enable_{family}_duo_attention_static_kv_cache_eval")
    ref = tiny(family, seed=1)
    model = copy.deepcopy(ref)
    heads = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    enable_static(model, heads.copy())
    ids = torch.randint(0, 211, (1, 700), generator=torch.Generator().manual_seed(2)).to(DEV)
    cache = mod.DuoAttentionStaticKVCache(model, heads, 1, 800, 512, 512)   # window covers the context
    chunks = [300, 257, 140, 1, 1, 1]
    pos = 0
    with torch.no_grad():
        for c in chunks:
            out = model(input_ids=ids[:, pos:pos + c], past_key_values=cache, use_cache=True)
            pos += c
            want = ref(input_ids=ids[:, :pos]).logits[:, -1:, :]
            assert out.logits.shape == want.shape
            assert _rel(out.logits, want) < 3e-2, (c, _rel(out.logits, want))   # bf16 models, 3 layers
    assert cache.kv_seq_len == 700


def test_graph_captured_generation_matches_eager(eager_decode_steps):
    """Whole patched HF model, greedy generation: DecodeStepGraph (one captured step, device-side cache
    lengths, the sampled token fed back inside the graph) produces the same tokens and logits as the eager
    loop, across the streaming window's fill -> slide transition."""
    from duo_attn.graph import DecodeStepGraph

    mod = __import__("duo_attn.patch.llama", fromlist=["x"])
    heads = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    ids = torch.randint(0, 211, (1, 40), generator=torch.Generator().manual_seed(3)).to(DEV)
    n_new = 12

    def run(use_graph):
        model = tiny("llama", seed=4)
        mod.enable_llama_duo_attention_static_kv_cache_eval(model, heads.copy())
        cache = mod.DuoAttentionStaticKVCache(model, heads, 1, 80, 8, 36)    # window 44: slides from step 5 on
        toks, logits = [], []
        with torch.no_grad():
            out = model(input_ids=ids, past_key_values=cache, use_cache=True)
            tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
            # one eager decode step first (loads every kernel / GEMM handle before any capture)
            out = model(input_ids=tok, past_key_values=cache, use_cache=True)
            toks.append(tok.clone()); logits.append(out.logits.clone())
            tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
            if use_graph:
                cur = tok.clone()
                state = {}

                def step():
                    o = model(input_ids=cur, past_key_values=cache, use_cache=True)
                    state["logits"] = o.logits
                    state["next"] = o.logits[:, -1, :].argmax(-1, keepdim=True)
                    return o.logits

                graph = DecodeStepGraph(cache, step, evict_after=0)
                for _ in range(n_new):
                    toks.append(cur.clone())
                    graph.replay()
                    logits.append(state["logits"].clone())
                    cur.copy_(state["next"])          # feed the sampled token back (static buffer)
            else:
                for _ in range(n_new):
                    toks.append(tok.clone())
                    out = model(input_ids=tok, past_key_values=cache, use_cache=True)
                    logits.append(out.logits.clone())
                    tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
        return torch.cat(toks, 1).cpu(), torch.cat(logits, 1).float().cpu(), cache.kv_seq_len

    t_e, l_e, n_e = run(False)
    t_g, l_g, n_g = run(True)
    assert n_e == n_g == 40 + 1 + n_new
    assert torch.equal(t_e, t_g)
    assert torch.equal(l_e, l_g)


def test_tuple_path_on_gpu_matches_hf_and_truncates():
    from duo_attn.patch import enable_duo_attention_eval

    ref = tiny("llama", seed=3)
    model = copy.deepcopy(ref)
    heads = np.array([[0.0, 1.0], [1.0, 0.0], [1.0, 1.0]])
    enable_duo_attention_eval(model, heads.copy(), 64, 1024)
    ids = torch.randint(0, 211, (1, 400), generator=torch.Generator().manual_seed(4)).to(DEV)
    past, pos = None, 0
    with torch.no_grad():
        for c in (256, 141, 1, 1, 1):
            out = model(input_ids=ids[:, pos:pos + c], past_key_values=past, use_cache=True)
            past = out.past_key_values
            pos += c
            want = ref(input_ids=ids[:, :pos]).logits[:, -1:, :]
            assert _rel(out.logits, want) < 3e-2
    assert past[0][0].shape == (2, 1, 400, 128) and past[0][1].shape == (2, 1, 400, 128)
    # a small window really evicts
    model2 = copy.deepcopy(ref)
    enable_duo_attention_eval(model2, heads.copy(), 16, 32)
    with torch.no_grad():
        out = model2(input_ids=ids[:, :300], use_cache=True)
        out = model2(input_ids=ids[:, 300:301], past_key_values=out.past_key_values, use_cache=True)
    assert out.past_key_values[0][1].shape == (2, 1, 48, 128)
