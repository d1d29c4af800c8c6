"""GPU: the reference's UNCHANGED decode loop is served by a HIP graph captured on the way, and a
batched decode step can be captured (item 7).

The reference loop (eval/efficiency/benchmark_static.py:96-105) is
    model(input_ids=pred, past_key_values=kv_cache, use_cache=True); kv_cache.evict_last(1)
per token, from Python.  ``duo_attn.graph.auto_decode_step`` captures that call after two eager steps and replays it:
bit-identical logits and cache contents to the eager steps, no host-to-device copy per token (``evict_last`` is mirrored on
the device counters with a launch), re-capture when the model's kernels change, eager for everything that is not that call.
"""
import copy

import numpy as np
import pytest
import torch

from helpers import ShapeModel, heads_from_counts
from test_golden_and_model_gpu import tiny

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _setup(family="llama", seed=21, heads=None, max_size=200, sink=16, recent=48):
    mod = __import__(f"duo_attn.patch.{family}", fromlist=["x"])
    model = tiny(family, seed=seed)
    heads = np.array([[1.0, 0.0], [0.0, 0.0], [1.0, 1.0]]) if heads is None else heads
    getattr(mod, f"enable_{family}_duo_attention_static_kv_cache_eval")(model, heads.copy())
    kv = mod.DuoAttentionStaticKVCache(model, heads, 1, max_size, sink, recent)
    return model, kv


def _pools(kv):
    return [t.clone() for lst in (kv.full_key_states_list, kv.full_value_states_list, kv.streaming_key_states_list,
                                  kv.streaming_value_states_list) for t in lst]


@pytest.mark.parametrize("family", ["llama", "mistral"])
@pytest.mark.parametrize("evict", [1, 0])
def test_reference_decode_loop_is_graph_replayed_and_bit_equal_to_eager(family, evict, monkeypatch):
    from duo_attn import graph

    ids = torch.randint(0, 211, (1, 70), generator=torch.Generator().manual_seed(1)).to(DEV)

    def loop(auto):
        monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", auto)
        model, kv = _setup(family)
        syncs = []
        orig = kv.sync_device_state
        logits = []
        kv.sync_device_state = lambda: (syncs.append(len(logits)), orig())[1]
        with torch.no_grad():
            out = model(input_ids=ids[:, :60], past_key_values=kv, use_cache=True)
            pred = out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
            for _ in range(12):
                out = model(input_ids=pred, past_key_values=kv, use_cache=True)      # the reference's call, verbatim
                logits.append(out.logits.clone())
                if evict:
                    kv.evict_last(1)
                else:
                    pred = out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
        torch.cuda.synchronize()
        return logits, kv, syncs

    l_e, kv_e, _ = loop(False)
    assert kv_e._decode_graph is None
    l_a, kv_a, syncs = loop(True)
    assert kv_a._decode_graph is not None, "the loop's third step should have captured the graph"
    for s, (a, b) in enumerate(zip(l_a, l_e)):
        assert a.shape == b.shape == (1, 1, 211) and torch.equal(a, b), f"step {s}"
    assert kv_a.kv_seq_len_list == kv_e.kv_seq_len_list and kv_a.streaming_kv_seq_len_list == kv_e.streaming_kv_seq_len_list
    for a, b in zip(_pools(kv_a), _pools(kv_e)):
        assert torch.equal(a, b)
    dev = kv_a.device_state.cpu()
    assert dev[:, 0].tolist() == kv_a.kv_seq_len_list and dev[:, 1].tolist() == kv_a.streaming_kv_seq_len_list
    # uploads only while the graph is built (during the loop's third step); evict_last is mirrored by a launch afterwards
    assert syncs and all(at == 2 for at in syncs), syncs
    # logits handed out earlier are the caller's: a later replay does not overwrite them
    assert not torch.equal(l_a[-1], l_a[-2]) or evict


def test_auto_graph_serves_the_next_prompt_and_recaptures_when_the_kernels_change(monkeypatch):
    from duo_attn import graph
    from duo_attn.patch import _duo

    ids = torch.randint(0, 211, (1, 120), generator=torch.Generator().manual_seed(2)).to(DEV)
    model, kv = _setup("llama", max_size=160)
    ref_model, ref_kv = _setup("llama", max_size=160)

    def decode(m, c, tok, n, auto):
        monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", auto)
        outs = []
        with torch.no_grad():
            for _ in range(n):
                o = m(input_ids=tok, past_key_values=c, use_cache=True)
                outs.append(o.logits.clone())
                tok = o.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
        return outs

    def prefill(m, c, lo, hi):
        with torch.no_grad():
            return m(input_ids=ids[:, lo:hi], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)

    t = prefill(model, kv, 0, 40)
    prefill(ref_model, ref_kv, 0, 40)
    for a, b in zip(decode(model, kv, t, 6, True), decode(ref_model, ref_kv, t, 6, False)):
        assert torch.equal(a, b)
    g1 = kv._decode_graph
    assert g1 is not None
    # next prompt, other length, prefilled eagerly in two chunks: same graph object, counters re-uploaded once
    for c in (kv, ref_kv):
        c.clear()
    for lo, hi in ((40, 95), (95, 118)):
        t = prefill(model, kv, lo, hi)
        prefill(ref_model, ref_kv, lo, hi)
    for a, b in zip(decode(model, kv, t, 5, True), decode(ref_model, ref_kv, t, 5, False)):
        assert torch.equal(a, b)
    assert kv._decode_graph is g1
    # the kernels behind the step change (module-by-module layers instead of the fused token-row linears): re-captured
    monkeypatch.setattr(_duo, "_FUSED_DECODE_LAYER", False)
    a5, b5 = decode(model, kv, t, 5, True), decode(ref_model, ref_kv, t, 5, False)
    for a, b in zip(a5, b5):
        assert torch.equal(a, b)
    assert kv._decode_graph is not None and kv._decode_graph is not g1
    for a, b in zip(_pools(kv), _pools(ref_kv)):
        assert torch.equal(a, b)


def test_calls_that_are_not_the_reference_decode_call_stay_eager(monkeypatch):
    from duo_attn import graph

    monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", True)
    ids = torch.randint(0, 211, (1, 50), generator=torch.Generator().manual_seed(3)).to(DEV)
    model, kv = _setup("llama")
    with torch.no_grad():
        model(input_ids=ids[:, :40], past_key_values=kv, use_cache=True)
        for t in range(40, 46):         # explicit position ids: not the reference's call
            pos = torch.tensor([[t]], device=DEV)
            model(input_ids=ids[:, t:t + 1], position_ids=pos, past_key_values=kv, use_cache=True)
        assert kv._decode_graph is None and getattr(kv, "_auto_graph", None) is None
        model(input_ids=ids[:, 46:48], past_key_values=kv, use_cache=True)          # two tokens at a time
        assert kv._decode_graph is None and getattr(kv, "_auto_graph", None) is None
    with torch.enable_grad():           # gradients enabled: never captured
        for t in range(48, 50):
            model(input_ids=ids[:, t:t + 1], past_key_values=kv, use_cache=True)
            model(input_ids=ids[:, t:t + 1], past_key_values=kv, use_cache=True)
            kv.evict_last(1)
    assert kv._decode_graph is None and getattr(kv, "_auto_graph", None) is None


@pytest.mark.parametrize("prefill,sink,recent", [(13, 4, 12), (650, 2, 300)])
@pytest.mark.parametrize("starts", [[0, 0], [5, 21]])
def test_batched_decode_step_is_capturable(starts, prefill, sink, recent):
    """B = 2 through the static attention core with device-side lengths (duo_decode_layer_batched_dev_bf16): one captured
    step replayed == the eager batched steps bit for bit, with equal and with per-row different RoPE positions (a row's
    offset from the cache length is fixed for the life of the sequence).  The 650-row variant (650 ... 657 rows: one count of 64-row units, so the eager steps choose the captured partition) has the retrieval heads split
    over several workgroups AND a saturated streaming pool whose update is folded into the scan — the combination in which
    every batch row once published its partials into row 0's workspace area (found by tests/fuzz_static_path.py)"""
    from duo_attn.graph import DecodeStepGraph
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    counts, Hq, Hkv, B = [1, 3, 0, 4], 16, 4, 2
    heads = heads_from_counts(counts, Hkv)

    def setup():
        g = torch.Generator().manual_seed(31)
        cache = DuoAttentionStaticKVCache(ShapeModel(len(counts), Hq, Hkv, D, device=DEV), heads, B, prefill + 51, sink, recent)
        mk = lambda S, h: torch.randn(B, S, h, D, generator=g).to(torch.bfloat16).to(DEV)
        for li in range(len(counts)):
            duo_static_attention_core(mk(prefill, Hq), mk(prefill, Hkv), mk(prefill, Hkv), cache, li, starts if starts[0] != starts[1] else starts[0],
                                      1.0, 1e4)
        qs, ks, vs = ([mk(1, h) for _ in counts] for h in (Hq, Hkv, Hkv))
        outs = [torch.zeros(B, 1, Hq, D, dtype=torch.bfloat16, device=DEV) for _ in counts]

        def step():
            past = cache.kv_seq_len
            pos = [s + past for s in starts]
            for li in range(len(counts)):
                outs[li].copy_(duo_static_attention_core(qs[li], ks[li], vs[li], cache, li,
                                                         pos if pos[0] != pos[1] else pos[0], 1.0, 1e4))
            return outs

        return cache, step, outs

    cache_e, step_e, outs_e = setup()
    eager = []
    for _ in range(7):
        step_e()
        eager.append([o.clone() for o in outs_e])
    cache_g, step_g, outs_g = setup()
    graph = DecodeStepGraph(cache_g, step_g, evict_after=0)
    for s in range(7):
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(outs_g, eager[s]):
            assert torch.equal(a, b), f"step {s}"
    assert cache_g.kv_seq_len_list == cache_e.kv_seq_len_list
    for a, b in zip(_pools(cache_g), _pools(cache_e)):
        assert torch.equal(a, b)
