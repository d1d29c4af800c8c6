"""BASELINE config[0] in miniature: the patch API on HuggingFace Llama / Mistral models on CPU
(host plumbing only; the oracle is plugged in as the device backend — the product itself has no CPU
path).  Property tests that need no golden data:
  (i)   all heads retrieval heads  => patched model == unpatched HF model;
  (ii)  weight reordering is invisible when sink+recent covers the whole context;
  (iii) chunked prefill + decode == single-shot for retrieval heads;
  (iv)  tuple path and static path agree.
fp32 models so the comparisons are tight.
"""
import copy

import numpy as np
import pytest
import torch
from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM


def tiny(family, seed=0):
    torch.manual_seed(seed)
    kw = dict(hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
              num_key_value_heads=2, head_dim=128, vocab_size=97, max_position_embeddings=2048,
              rope_theta=10000.0, attn_implementation="eager", tie_word_embeddings=False)
    if family == "llama":
        model = LlamaForCausalLM(LlamaConfig(**kw))
    else:
        model = MistralForCausalLM(MistralConfig(sliding_window=None, **kw))
    return model.float().eval()


@pytest.fixture
def exact_backend():
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    yield
    backend._set_backend_for_testing(None)


def hf_last_logits(model, ids):
    with torch.no_grad():
        return model(input_ids=ids).logits[:, -1:, :]


def run_chunks(model, ids, chunks, past=None):
    outs = []
    pos = 0
    with torch.no_grad():
        for c in chunks:
            out = model(input_ids=ids[:, pos:pos + c], past_key_values=past, use_cache=True)
            past = out.past_key_values
            outs.append(out.logits)
            pos += c
    return outs, past


@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_tuple_path_all_full_equals_hf(family, exact_backend):
    from duo_attn.patch import enable_duo_attention_eval, get_full_attention_heads

    ref = tiny(family)
    model = copy.deepcopy(ref)
    heads = np.ones((2, 2))
    enable_duo_attention_eval(model, heads, 4, 8)
    ids = torch.randint(0, 97, (1, 33), generator=torch.Generator().manual_seed(1))
    outs, past = run_chunks(model, ids, [20, 9, 1, 1, 1, 1])
    for n, lg in zip(np.cumsum([20, 9, 1, 1, 1, 1]), outs):
        assert lg.shape == (1, 1, 97) and lg.dtype == torch.float32
        torch.testing.assert_close(lg, hf_last_logits(ref, ids[:, :n]), rtol=2e-4, atol=2e-4)
    # tuple cache layout: (full_KV [2B, nf, N, D], streaming_KV [2B, ns, n, D]) per layer
    assert len(past) == 2 and past[0][0].shape == (2, 2, 33, 128) and past[0][1].shape == (2, 0, 12, 128)
    assert [h.tolist() for h in get_full_attention_heads(model)] == [[1.0, 1.0], [1.0, 1.0]]


@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_reordering_is_invisible_when_window_covers_context(family, exact_backend):
    """mixed retrieval/streaming heads with sink+recent >= context: streaming heads see everything, so any
    difference from HF can only come from a wrong q/k/v/o permutation or a wrong head split."""
    from duo_attn.patch import enable_duo_attention_eval

    ref = tiny(family, seed=3)
    model = copy.deepcopy(ref)
    heads = np.array([[0.0, 1.0], [1.0, 0.0]])    # layer 0: head 1 retrieval; layer 1: head 0 retrieval
    enable_duo_attention_eval(model, heads, 16, 64)
    ids = torch.randint(0, 97, (1, 40), generator=torch.Generator().manual_seed(2))
    outs, past = run_chunks(model, ids, [25, 10, 1, 1, 1, 1, 1])
    for n, lg in zip(np.cumsum([25, 10, 1, 1, 1, 1, 1]), outs):
        torch.testing.assert_close(lg, hf_last_logits(ref, ids[:, :n]), rtol=2e-4, atol=2e-4)
    assert past[0][0].shape == (2, 1, 40, 128) and past[0][1].shape == (2, 1, 40, 128)


def test_tuple_cache_grows_in_place_and_survives_branching(exact_backend):
    """The retrieval part of the tuple cache is a view of a growing arena: linear generation appends in
    place (no O(N) re-concatenation per token), and a caller that goes back to an OLDER tuple and continues
    differently still gets the right answer (the stale view is copied into a fresh arena)."""
    from duo_attn.patch import enable_duo_attention_eval

    ref = tiny("llama", seed=6)
    model = copy.deepcopy(ref)
    enable_duo_attention_eval(model, np.ones((2, 2)), 4, 8)
    ids = torch.randint(0, 97, (1, 30), generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        out = model(input_ids=ids[:, :20], use_cache=True)
        past20 = out.past_key_values
        base_ptr = past20[0][0].data_ptr()
        o21 = model(input_ids=ids[:, 20:21], past_key_values=past20, use_cache=True)
        o22 = model(input_ids=ids[:, 21:22], past_key_values=o21.past_key_values, use_cache=True)
        # in place: same storage, longer view; the older tuples are still intact views
        assert o22.past_key_values[0][0].data_ptr() == base_ptr and o22.past_key_values[0][0].shape[2] == 22
        assert past20[0][0].shape[2] == 20 and o21.past_key_values[0][0].shape[2] == 21
        torch.testing.assert_close(o22.logits, hf_last_logits(ref, ids[:, :22]), rtol=2e-4, atol=2e-4)
        # branch: continue from the 20-token cache with a DIFFERENT token
        alt = (ids[:, 20:21] + 1) % 97
        b21 = model(input_ids=alt, past_key_values=past20, use_cache=True)
        want = hf_last_logits(ref, torch.cat([ids[:, :20], alt], 1))
        torch.testing.assert_close(b21.logits, want, rtol=2e-4, atol=2e-4)
        # ... and the first branch can still be continued from its own latest tuple
        o23 = model(input_ids=ids[:, 22:23], past_key_values=o22.past_key_values, use_cache=True)
        torch.testing.assert_close(o23.logits, hf_last_logits(ref, ids[:, :23]), rtol=2e-4, atol=2e-4)


def test_streaming_heads_truncate_and_change_logits(exact_backend):
    from duo_attn.patch import enable_duo_attention_eval

    ref = tiny("llama", seed=4)
    model = copy.deepcopy(ref)
    enable_duo_attention_eval(model, np.array([[0.0, 1.0], [1.0, 0.0]]), 2, 6)
    ids = torch.randint(0, 97, (1, 30), generator=torch.Generator().manual_seed(5))
    outs, past = run_chunks(model, ids, [12, 12, 1, 1, 1, 1, 1, 1])
    assert past[0][1].shape == (2, 1, 8, 128)      # sink 2 + recent 6
    assert past[0][0].shape == (2, 1, 30, 128)
    assert not torch.allclose(outs[-1], hf_last_logits(ref, ids), atol=1e-3)


@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_static_path_all_full_equals_hf_and_tuple_path(family, exact_backend):
    mod = __import__(f"duo_attn.patch.{family}", fromlist=["x"])
    enable_static = getattr(mod, f"enable_{family}_duo_attention_static_kv_cache_eval")
    Cache = mod.DuoAttentionStaticKVCache

    ref = tiny(family, seed=6)
    model = copy.deepcopy(ref)
    heads = np.ones((2, 2))
    enable_static(model, heads)
    ids = torch.randint(0, 97, (1, 30), generator=torch.Generator().manual_seed(7))
    cache = Cache(model, heads, 1, 64, 4, 8)
    outs, past = run_chunks(model, ids, [17, 10, 1, 1, 1], past=cache)
    assert past is cache and cache.kv_seq_len == 30
    for n, lg in zip(np.cumsum([17, 10, 1, 1, 1]), outs):
        assert lg.shape == (1, 1, 97)
        # static path: fp32 RoPE from (theta, factor) instead of HF's cos/sin tables -> tiny differences
        torch.testing.assert_close(lg, hf_last_logits(ref, ids[:, :n]), rtol=1e-3, atol=1e-3)


def test_static_path_mixed_heads_matches_tuple_path(exact_backend):
    """same pattern through both cache formats (chunk sizes identical: streaming-head outputs depend on
    the chunking, reference llama.py:385-412)."""
    from duo_attn.patch import enable_duo_attention_eval
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    base = tiny("llama", seed=8)
    heads = np.array([[1.0, 0.0], [0.0, 1.0]])
    m_tuple, m_static = copy.deepcopy(base), copy.deepcopy(base)
    enable_duo_attention_eval(m_tuple, heads.copy(), 3, 5)
    enable_llama_duo_attention_static_kv_cache_eval(m_static, heads.copy())
    ids = torch.randint(0, 97, (1, 41), generator=torch.Generator().manual_seed(9))
    chunks = [16, 16, 5, 1, 1, 1, 1]
    o_t, _ = run_chunks(m_tuple, ids, chunks)
    cache = DuoAttentionStaticKVCache(m_static, heads, 1, 50, 3, 5)
    o_s, _ = run_chunks(m_static, ids, chunks, past=cache)
    for a, b in zip(o_t, o_s):
        torch.testing.assert_close(a, b.float(), rtol=1e-3, atol=1e-3)
    assert cache.streaming_kv_seq_len == 8 and cache.kv_seq_len == 41


def test_full_attention_tuple_baseline(exact_backend):
    from duo_attn.patch.tuple_kv_cache import enable_tuple_kv_cache

    ref = tiny("mistral", seed=10)
    model = copy.deepcopy(ref)
    enable_tuple_kv_cache(model)
    ids = torch.randint(0, 97, (1, 21), generator=torch.Generator().manual_seed(11))
    outs, past = run_chunks(model, ids, [13, 6, 1, 1])
    for n, lg in zip(np.cumsum([13, 6, 1, 1]), outs):
        torch.testing.assert_close(lg, hf_last_logits(ref, ids[:, :n]), rtol=2e-4, atol=2e-4)
    assert past[1][0].shape == (1, 2, 21, 128)


def test_without_a_backend_the_patched_model_refuses_cpu():
    """no oracle plugged in: the product path must fail loudly on CPU tensors"""
    from duo_attn import _hip, backend
    from duo_attn.patch import enable_duo_attention_eval

    backend._set_backend_for_testing(None)
    model = tiny("llama")
    model = model.to(torch.bfloat16)
    enable_duo_attention_eval(model, np.ones((2, 2)), 4, 8)
    with pytest.raises(_hip.DuoHipError, match="no CPU fallback"):
        model(input_ids=torch.zeros(1, 4, dtype=torch.long), use_cache=True)


def test_config0_llama2_7b_shape_4k_prompt(exact_backend):
    """BASELINE configs[0] at its STATED size: Llama-2-7B-32K shape (hidden 4096, 32 q = kv heads of 128 dims, linear
    rope scaling factor 8), a 4096-token prompt through enable_duo_attention_eval / the static path on the CPU
    (oracle as backend).  One decoder layer of that shape (the 32 layers are identical in shape; a 7B model is 27 GB
    in fp32), pattern with 8 of 32 retrieval heads (sparsity 0.75).
      * static path, window covering the prompt, chunks of 2048 + 3 decode tokens == unpatched HF eager logits;
      * the same pattern with the shipped sink 128 + recent 256: cache geometry of the 4K prompt."""
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    torch.manual_seed(5)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=1024, num_hidden_layers=1, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=211, max_position_embeddings=32768, rope_theta=10000.0,
                      rope_scaling={"rope_type": "linear", "factor": 8.0}, attn_implementation="eager",
                      tie_word_embeddings=False)
    ref = LlamaForCausalLM(cfg).float().eval()
    N = 4096
    ids = torch.randint(0, 211, (1, N + 3), generator=torch.Generator().manual_seed(6))
    heads = np.zeros((1, 32))
    heads[0, [1, 4, 5, 9, 16, 22, 27, 31]] = 1.0
    model = copy.deepcopy(ref)
    enable_llama_duo_attention_static_kv_cache_eval(model, heads.copy())
    kv = DuoAttentionStaticKVCache(model, heads, 1, N + 8, 2048, 2560)        # window >= context: == full attention
    outs, _ = run_chunks(model, ids, [2048, 2048, 1, 1, 1], past=kv)
    want = hf_last_logits(ref, ids)
    assert kv.kv_seq_len == N + 3 and kv.num_full_kv_head_list == [8]
    rel = ((outs[-1] - want).norm() / want.norm()).item()
    assert rel < 2e-3, rel          # fp32 model; linear rope factor 8 applied identically on both sides
    # shipped window: the streaming pool saturates at sink + recent rows, the retrieval pool holds the prompt
    kv2 = DuoAttentionStaticKVCache(model, heads, 1, N + 8, 128, 256)
    outs2, _ = run_chunks(model, ids[:, :N], [2048, 2048], past=kv2)
    assert kv2.kv_seq_len == N and kv2.streaming_kv_seq_len == 384
    assert kv2.full_key_states_list[0].shape == (1, N + 8, 8, 128) and kv2.streaming_key_states_list[0].shape == (1, 384, 24, 128)
    assert kv2.memory_usage == 2 * ((N + 8) * 8 + 384 * 24) * 128 * 4              # K and V, fp32 model here
    assert torch.isfinite(outs2[-1]).all() and not torch.allclose(outs2[-1], outs[1], atol=1e-3)   # eviction changes the result


def test_int4_cache_is_pipeline_stage_aware():
    """on a layer-pipeline stage the INT4 cache must hold the pools of the stage's OWN layers (indexed by
    local layer number, like DuoAttentionStaticKVCache), not of global layers 0..k."""
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache
    from duo_attn.pipeline import LayerPipeline, PPState
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache
    from helpers import ShapeModel

    heads = [[1, 0, 0, 0], [1, 1, 0, 0], [0, 0, 0, 0], [1, 1, 1, 0], [1, 1, 1, 1]]
    model = ShapeModel(5, 8, 4, 128, dtype=torch.float16)
    pipe = LayerPipeline(5, rank=1, world_size=2)                    # even split: stage 1 = layers [3, 5) or [2, 5)
    model._duo_pp = PPState(pipe, "cpu", 5)
    lo, hi = pipe.first_layer, pipe.last_layer
    want = [int(sum(h)) for h in heads[lo:hi]]
    for rows in (heads, heads[lo:hi]):                                 # whole-model pattern or the stage's rows
        kv = DuoAttentionStaticINT4KVCache(model, rows, 1, 32, 4, 8, 16)
        ref = DuoAttentionStaticKVCache(model, rows, 1, 32, 4, 8)
        assert kv.num_layers == ref.num_layers == hi - lo
        assert kv.num_full_kv_head_list == ref.num_full_kv_head_list == want
        assert len(kv.kv_seq_len_list) == hi - lo and kv.kv_seq_len == 0
        assert [c.quantized_data.shape[2] for c in kv.full_key_caches] == want
    with pytest.raises(ValueError):
        DuoAttentionStaticINT4KVCache(model, heads[:4], 1, 32, 4, 8, 16)
