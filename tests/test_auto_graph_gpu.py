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

from helpers import ShapeModel, heads_from_counts, rel_close
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
    # uploads only while a graph is built: during the loop's third step, and — the growing loop (60 -> 72 rows) leaves the
    # 64-row length bucket at its fifth step — during the re-capture there; evict_last is mirrored by a launch afterwards
    assert syncs and set(syncs) == ({2} if evict else {2, 4}), syncs
    assert kv_a._decode_graph.captures == (1 if evict else 2)
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


# ------------------------------------------------------------------------------------------------------------------
# the captured grid follows the context (length buckets), the signature covers every weight, hooks keep
# firing, every graph owns its partials
# ------------------------------------------------------------------------------------------------------------------
def _decode(model, kv, tok, n, auto, monkeypatch, feed=None):
    from duo_attn import graph

    monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", auto)
    outs = []
    with torch.no_grad():
        for i in range(n):
            o = model(input_ids=tok, past_key_values=kv, use_cache=True)
            outs.append(o.logits.clone())
            tok = o.logits[:, -1, :].argmax(dim=-1).unsqueeze(1) if feed is None else feed[:, i:i + 1]
    return outs


def test_plan_bucket_is_the_power_of_two_of_the_64_token_units():
    from duo_attn import _hip

    b = _hip.load_library().duo_decode_plan_bucket
    assert [b(n) for n in (-3, 0, 1, 64, 65, 128, 129, 256, 257, 131072, 131073, 3300000)] == \
        [0, 0, 1, 1, 2, 2, 4, 4, 8, 2048, 4096, 65536]


def test_the_step_is_captured_again_when_the_context_leaves_its_length_bucket(monkeypatch):
    """A generation that grows through two bucket boundaries (64 and 128 rows), then ``clear()`` and a much shorter prompt
    through the same cache: the step is re-captured each time the library would plan another split-KV grid
    (``DecodeStepGraph.captures``), and EVERY step — before, at and after a boundary — is bit-equal to the eager loop, whose
    launches are planned from the same bucket.  (Before: a graph captured at a short context kept its grid for good.)"""
    from duo_attn import graph

    ids = torch.randint(0, 211, (1, 400), generator=torch.Generator().manual_seed(7)).to(DEV)
    model, kv = _setup("llama", max_size=300)
    ref_model, ref_kv = _setup("llama", max_size=300)
    with torch.no_grad():
        for m, c in ((model, kv), (ref_model, ref_kv)):
            t = m(input_ids=ids[:, :58], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
    feed = ids[:, 100:190]                                       # teacher-forced: both loops see the same tokens
    got = _decode(model, kv, t, 90, True, monkeypatch, feed)
    want = _decode(ref_model, ref_kv, t, 90, False, monkeypatch, feed)
    for s, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"step {s} (cache length {58 + s})"
    g = kv._decode_graph
    assert g is not None and ref_kv._decode_graph is None
    assert g.captures == 3, g.captures                           # at 60 rows, past 64 rows, past 128 rows
    assert g.plan_key == graph.plan_key(kv) == (4, 2)            # 148 rows -> 3 units -> 4; window 64 + 1 rows -> 2
    for c in (kv, ref_kv):
        c.clear()
    with torch.no_grad():
        for m, c in ((model, kv), (ref_model, ref_kv)):
            t = m(input_ids=ids[:, 200:220], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
    got = _decode(model, kv, t, 5, True, monkeypatch)
    want = _decode(ref_model, ref_kv, t, 5, False, monkeypatch)
    for s, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"second prompt, step {s}"
    assert kv._decode_graph is g and g.captures == 4 and g.plan_key == (1, 1)
    for a, b in zip(_pools(kv), _pools(ref_kv)):
        assert torch.equal(a, b)


def test_llama3_geometry_long_context_the_default_decode_path_equals_eager(monkeypatch):
    """Llama-3-8B head geometry (32 q / 8 kv heads, two layers), 40 000-token context prefilled in 8 192-token chunks, the
    reference's decode loop with ``evict_last(1)``: the path users get by default (automatic graph) against the eager loop,
    bit for bit — first captured at a 100-token context, then serving the long one through the same cache (the capture
    follows the bucket: this is the case where the short context's grid would scan a 40 000-row pool with one workgroup per
    kv head)."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from duo_attn import graph
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    torch.manual_seed(5)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=128, max_position_embeddings=1048576, rope_theta=3580165449.0,
                      attn_implementation="eager", tie_word_embeddings=False)
    heads = np.array([[1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0], [1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0, 0.0]])
    N, C = 40000, 8192
    ids = torch.randint(0, 128, (1, N + 100), generator=torch.Generator().manual_seed(6)).to(DEV)

    def run(auto):
        monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", auto)
        torch.manual_seed(5)
        model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval().to(DEV)
        enable_llama_duo_attention_static_kv_cache_eval(model, heads.copy())
        kv = DuoAttentionStaticKVCache(model, heads, 1, N + 16, 128, 256)
        logits = []
        with torch.no_grad():
            pred = model(input_ids=ids[:, N:N + 100], past_key_values=kv, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
            for _ in range(4):                                   # a short context first
                logits.append(model(input_ids=pred, past_key_values=kv, use_cache=True).logits.clone())
                kv.evict_last(1)
            short_key = graph.plan_key(kv)
            kv.clear()
            for lo in range(0, N, C):
                out = model(input_ids=ids[:, lo:min(lo + C, N)], past_key_values=kv, use_cache=True)
            pred = out.logits[:, -1, :].argmax(-1).unsqueeze(1)
            for _ in range(6):                                   # the reference's loop, benchmark_static.py:96-105
                logits.append(model(input_ids=pred, past_key_values=kv, use_cache=True).logits.clone())
                kv.evict_last(1)
        torch.cuda.synchronize()
        return logits, kv, short_key

    l_a, kv_a, short_key = run(True)
    l_e, kv_e, _ = run(False)
    assert kv_e._decode_graph is None and kv_a._decode_graph is not None
    assert short_key == (2, 2) and kv_a._decode_graph.plan_key == (1024, 8) and kv_a._decode_graph.captures == 2
    for s, (a, b) in enumerate(zip(l_a, l_e)):
        assert torch.isfinite(a).all() and torch.equal(a, b), f"step {s}"
    for a, b in zip(_pools(kv_a), _pools(kv_e)):
        assert torch.equal(a, b)


def test_a_partial_weight_swap_retires_the_captured_step(monkeypatch):
    """the signature covers the storage of EVERY parameter the step reads, not only q_proj / down_proj: swapping the ``.data``
    of one k_proj, one norm and the lm_head (new storage, new values) after the capture re-captures the step — a replay of the
    old graph would read the old (freed) storage — and the result equals the eager loop on an identically modified model"""
    ids = torch.randint(0, 211, (1, 60), generator=torch.Generator().manual_seed(8)).to(DEV)
    model, kv = _setup("mistral")
    ref_model, ref_kv = _setup("mistral")
    with torch.no_grad():
        for m, c in ((model, kv), (ref_model, ref_kv)):
            t = m(input_ids=ids[:, :50], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
    for a, b in zip(_decode(model, kv, t, 5, True, monkeypatch), _decode(ref_model, ref_kv, t, 5, False, monkeypatch)):
        assert torch.equal(a, b)
    g1 = kv._decode_graph
    assert g1 is not None
    gen = torch.Generator().manual_seed(9)
    for pick in (lambda m: m.model.layers[1].self_attn.k_proj.weight, lambda m: m.model.layers[2].post_attention_layernorm.weight,
                 lambda m: m.lm_head.weight):
        new = (torch.randn(pick(model).shape, generator=gen) * 0.05).to(torch.bfloat16).to(DEV)
        for m in (model, ref_model):
            pick(m).data = new.clone()
    a5, b5 = _decode(model, kv, t, 5, True, monkeypatch), _decode(ref_model, ref_kv, t, 5, False, monkeypatch)
    assert kv._decode_graph is not None and kv._decode_graph is not g1, "the swap did not retire the captured step"
    for s, (a, b) in enumerate(zip(a5, b5)):
        assert torch.equal(a, b), f"step {s} after the swap"
    # a REPLACED module (new object, new parameter): the module tree is walked per call, not cached
    g2 = kv._decode_graph
    w = (torch.randn(model.lm_head.weight.shape, generator=gen) * 0.05).to(torch.bfloat16).to(DEV)
    for m in (model, ref_model):
        head = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, dtype=torch.bfloat16, device=DEV)
        head.weight.data = w.clone()
        m.lm_head = head
    a5, b5 = _decode(model, kv, t, 4, True, monkeypatch), _decode(ref_model, ref_kv, t, 4, False, monkeypatch)
    assert kv._decode_graph is not None and kv._decode_graph is not g2, "the replaced lm_head did not retire the captured step"
    for s, (a, b) in enumerate(zip(a5, b5)):
        assert torch.equal(a, b), f"step {s} after the module replacement"


def test_forward_hooks_keep_firing(monkeypatch):
    """a replay does not re-enter Python, so a model with a forward (pre-)hook on any module is decoded eagerly — the hook
    sees every step; once it is removed the loop is captured"""
    ids = torch.randint(0, 211, (1, 50), generator=torch.Generator().manual_seed(10)).to(DEV)
    model, kv = _setup("llama")
    ref_model, ref_kv = _setup("llama")
    fired = []
    # (the MLP module is one the package's forwards go AROUND — SwiGLU product fused on chunks, the whole layer fused on
    #  decode steps: with a hook on it they call the module instead, as the reference does)
    h = model.model.layers[1].mlp.register_forward_hook(lambda mod, a, out: fired.append(1))
    with torch.no_grad():
        t = model(input_ids=ids[:, :40], past_key_values=kv, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
        ref_model(input_ids=ids[:, :40], past_key_values=ref_kv, use_cache=True)
    assert len(fired) == 1
    got = _decode(model, kv, t, 6, True, monkeypatch, ids[:, 40:46])         # (teacher-forced: both models see the same tokens)
    assert len(fired) == 1 + 6 and kv._decode_graph is None
    h.remove()
    got += _decode(model, kv, t, 6, True, monkeypatch, ids[:, 44:50])
    assert len(fired) == 7 and kv._decode_graph is not None
    # one layer ran module by module while hooked: logits within the projections' summation-order noise of the unhooked model
    want = _decode(ref_model, ref_kv, t, 6, False, monkeypatch, ids[:, 40:46]) + _decode(ref_model, ref_kv, t, 6, False, monkeypatch, ids[:, 44:50])
    for a, b in zip(got, want):
        rel_close(a, b, 8e-3, "auto-graph: one hooked layer module by module vs the fused model, logits")     # (measured: 1.7e-3 ... 3.6e-3)


def test_forward_hooks_fire_on_the_tuple_path_too():
    from duo_attn.patch import enable_duo_attention_eval

    model = tiny("mistral", seed=17)
    enable_duo_attention_eval(model, np.array([[0.0, 1.0], [1.0, 1.0], [0.0, 0.0]]), 16, 48)
    ids = torch.randint(0, 211, (1, 40), generator=torch.Generator().manual_seed(18)).to(DEV)
    fired = {"mlp": 0, "norm": 0, "o": 0}
    hs = [model.model.layers[0].mlp.register_forward_hook(lambda *a: fired.__setitem__("mlp", fired["mlp"] + 1)),
          model.model.layers[1].input_layernorm.register_forward_pre_hook(lambda *a: fired.__setitem__("norm", fired["norm"] + 1)),
          model.model.layers[2].self_attn.o_proj.register_forward_hook(lambda *a: fired.__setitem__("o", fired["o"] + 1))]
    with torch.no_grad():
        past = model(input_ids=ids[:, :30], past_key_values=None, use_cache=True).past_key_values
        for t in range(30, 35):
            past = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True).past_key_values
    assert fired == {"mlp": 6, "norm": 6, "o": 6}, fired
    for h in hs:
        h.remove()


def test_every_captured_step_owns_its_split_kv_partials(monkeypatch):
    """every capture runs on torch's one shared capture stream: the graphs of two caches must not bake the same partials
    buffer into their launches.  Two models, two caches, both captured: disjoint scratch, and replays issued alternately on
    two streams equal the eager loops.  (The replays are NOT overlapped here: a whole captured model step also contains the
    GEMM library's launches, whose workspace torch keeps per capture stream — whether that tolerates concurrent replays is
    not this package's to promise; what it owns, the split-KV partials and tickets, is per graph.)"""
    ids = torch.randint(0, 211, (1, 700), generator=torch.Generator().manual_seed(12)).to(DEV)
    pairs, refs = [_setup("llama", seed=31 + i, max_size=720) for i in range(2)], [_setup("llama", seed=31 + i, max_size=720) for i in range(2)]
    toks = []
    with torch.no_grad():
        for (m, c), (rm, rc) in zip(pairs, refs):
            toks.append(m(input_ids=ids[:, :650], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1))
            rm(input_ids=ids[:, :650], past_key_values=rc, use_cache=True)
    for (m, c), t in zip(pairs, toks):          # 650 rows: the retrieval heads are split, the partials buffer is in use
        _decode(m, c, t, 3, True, monkeypatch, ids[:, 650:653])
    g = [c._decode_graph for _, c in pairs]
    assert all(x is not None for x in g)
    bufs = [[t for t in x._scratch.values()] for x in g]
    assert bufs[0] and bufs[1]
    assert not {t.data_ptr() for t in bufs[0]} & {t.data_ptr() for t in bufs[1]}
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    torch.cuda.synchronize()
    from duo_attn import graph
    monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", True)
    with torch.no_grad():
        for s in range(8):
            for i, ((m, c), st) in enumerate(zip(pairs, streams)):
                with torch.cuda.stream(st):
                    got[i].append(m(input_ids=ids[:, 652 + s:653 + s], past_key_values=c, use_cache=True).logits)
                st.synchronize()
    torch.cuda.synchronize()
    for i, ((rm, rc), t) in enumerate(zip(refs, toks)):
        want = _decode(rm, rc, t, 11, False, monkeypatch, ids[:, 650:661])[3:]
        for s, (a, b) in enumerate(zip(got[i], want)):
            assert torch.equal(a, b), f"model {i}, step {s}"


def test_a_failed_recapture_falls_back_to_eager_steps(monkeypatch):
    """the first capture works, the one for the next length bucket raises (made to, here): nothing was launched, the host
    counters are restored, the loop goes on eagerly with a warning — and still equals the eager loop bit for bit"""
    from duo_attn import graph

    ids = torch.randint(0, 211, (1, 100), generator=torch.Generator().manual_seed(19)).to(DEV)
    model, kv = _setup("llama")
    ref_model, ref_kv = _setup("llama")
    with torch.no_grad():
        for m, c in ((model, kv), (ref_model, ref_kv)):
            t = m(input_ids=ids[:, :60], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
    orig, n = graph.DecodeStepGraph._capture, []

    def flaky(self):
        n.append(1)
        if len(n) > 1:
            raise RuntimeError("no memory for the capture (simulated)")
        return orig(self)

    monkeypatch.setattr(graph.DecodeStepGraph, "_capture", flaky)
    with pytest.warns(UserWarning, match="re-capturing the decode step"):
        got = _decode(model, kv, t, 12, True, monkeypatch, ids[:, 60:72])          # 60 -> 72 rows: leaves the 64-row bucket at step 5
    want = _decode(ref_model, ref_kv, t, 12, False, monkeypatch, ids[:, 60:72])
    assert len(n) == 2 and kv._decode_graph is None and kv._auto_graph_failed
    for s, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"step {s}"
    assert kv.kv_seq_len_list == ref_kv.kv_seq_len_list
    for a, b in zip(_pools(kv), _pools(ref_kv)):
        assert torch.equal(a, b)


def test_a_direct_graph_user_can_retry_after_a_failed_recapture():
    """``_capture`` assigns graph / plan_key / output only after the capture succeeded.  A direct
    ``DecodeStepGraph`` user who catches ``RecaptureError`` keeps the previous output tensor, finds ``plan_key`` reset (so the
    next ``replay()`` captures again instead of replaying an un-captured graph object), and the retried capture equals the
    eager loop bit for bit."""
    from duo_attn.graph import DecodeStepGraph, RecaptureError

    ids = torch.randint(0, 211, (1, 100), generator=torch.Generator().manual_seed(23)).to(DEV)
    model, kv = _setup("llama")
    ref_model, ref_kv = _setup("llama")
    with torch.no_grad():
        for m, c in ((model, kv), (ref_model, ref_kv)):
            t = m(input_ids=ids[:, :62], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
        for _ in range(2):          # warm: kernels, GEMM handles
            model(input_ids=t, past_key_values=kv, use_cache=True)
            kv.evict_last(1)
    tok = t.clone()

    def step():
        with torch.no_grad():
            return model(input_ids=tok, past_key_values=kv, use_cache=True, _duo_no_auto_graph=True).logits

    g = DecodeStepGraph(kv, step, evict_after=0)
    first_graph, first_out = g.graph, g.output
    got = [g.replay().clone() for _ in range(2)]            # 62 -> 64 rows: still the first bucket
    orig, fail = g._body, [True]

    def body():
        if fail[0]:
            fail[0] = False
            raise RuntimeError("simulated capture failure")
        return orig()

    g._body = body
    lens = list(kv.kv_seq_len_list)
    with pytest.raises(RecaptureError):
        g.replay()                                          # 65 rows: the next bucket; its capture fails
    assert g.plan_key is None and g.graph is first_graph and g.output is first_out and g.captures == 1
    assert kv.kv_seq_len_list == lens and not kv.use_device_state
    got += [g.replay().clone() for _ in range(3)]           # retried: captured, replayed
    assert g.captures == 2 and g.plan_key is not None and g.graph is not first_graph
    with torch.no_grad():
        want = [ref_model(input_ids=t, past_key_values=ref_kv, use_cache=True).logits for _ in range(5)]
    for s, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"step {s}"


def test_retired_graphs_are_released_without_a_device_wide_wait():
    """a dropped DecodeStepGraph hands its graph object to ``graph._retired`` with the event of its last replay; the list
    drains once that event has completed — and never from inside a stream capture"""
    from duo_attn import graph
    from duo_attn.graph import DecodeStepGraph

    ids = torch.randint(0, 211, (1, 100), generator=torch.Generator().manual_seed(29)).to(DEV)
    model, kv = _setup("llama")
    with torch.no_grad():
        t = model(input_ids=ids[:, :40], past_key_values=kv, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)
        model(input_ids=t, past_key_values=kv, use_cache=True)
        kv.evict_last(1)

    def step():
        with torch.no_grad():
            return model(input_ids=t, past_key_values=kv, use_cache=True, _duo_no_auto_graph=True).logits

    import gc

    gc.collect()                    # (captured steps of earlier tests that only the cyclic collector can reach: gone now)
    torch.cuda.synchronize()
    graph._drain_retired(block=True)
    assert graph._retired == []
    g = DecodeStepGraph(kv, step, evict_after=1)
    for _ in range(3):
        g.replay()
    side = torch.cuda.Stream()
    other = torch.cuda.CUDAGraph()
    x = torch.zeros(8, device=DEV)
    with torch.cuda.graph(other, stream=side):
        del g                       # destructor inside somebody else's capture: retire only, no query, no wait
        gc.collect()
        x += 1
    assert len(graph._retired) == 1
    other.replay()
    torch.cuda.synchronize()
    assert x.sum().item() == 8      # the foreign capture survived the destructor
    graph._drain_retired()
    assert graph._retired == []
