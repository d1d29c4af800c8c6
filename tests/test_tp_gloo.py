"""Head-parallel tensor parallelism (duo_attn/tp.py) on CPU: world size 2, gloo, oracle as device backend.
The sharded model (balanced retrieval-head assignment, column/row-sliced projections, two all-reduces per
layer) must reproduce the single-process patched model: chunked prefill + decode through the static cache."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths():
    for p in (ROOT, os.path.join(ROOT, "duo-attention_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _tiny():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=3, num_attention_heads=8,
                      num_key_value_heads=4, head_dim=128, vocab_size=97, max_position_embeddings=2048,
                      rope_theta=10000.0, attn_implementation="eager", tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).float().eval()


HEADS = np.array([[1.0, 1.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, 1.0]])
CHUNKS = [23, 17, 1, 1]


def _run(model, heads, max_size=64, wrap=None):
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    enable_llama_duo_attention_static_kv_cache_eval(model, heads.copy())
    if wrap is not None:
        wrap(model)
    cache = DuoAttentionStaticKVCache(model, heads, 1, max_size, 4, 8)
    ids = torch.randint(0, 97, (1, sum(CHUNKS)), generator=torch.Generator().manual_seed(1))
    outs, pos = [], 0
    with torch.no_grad():
        for c in CHUNKS:
            outs.append(model(input_ids=ids[:, pos:pos + c], past_key_values=cache, use_cache=True).logits)
            pos += c
    return torch.cat(outs, 1)


def _run_tuple(model):
    """chunked prefill + decode through the tuple cache of a model patched with enable_duo_attention_eval"""
    ids = torch.randint(0, 97, (1, sum(CHUNKS)), generator=torch.Generator().manual_seed(1))
    outs, pos, past = [], 0, None
    with torch.no_grad():
        for c in CHUNKS:
            o = model(input_ids=ids[:, pos:pos + c], past_key_values=past, use_cache=True)
            past = o.past_key_values
            outs.append(o.logits)
            pos += c
    return torch.cat(outs, 1)


def _worker(rank, world, port, q, mode="explicit"):
    _paths()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from duo_attn import backend
        from duo_attn.tp import shard_model_for_tp
        from oracle.duo_oracle import OracleBackend

        backend._set_backend_for_testing(OracleBackend(round_p=False))
        model = _tiny()
        if mode == "fused":
            # the decode steps' layers in the fused form (duo_decode_layer_fused): on a shard o_proj / down_proj are the
            # local product -> all-reduce -> residual add (_out_linear); driven on the CPU through the oracle's token_linear_ref
            from duo_attn.patch import _duo
            from test_token_linear_cpu import _FusedOracleBackend, _with_fused_layers

            be = _FusedOracleBackend()
            backend._set_backend_for_testing(be)
            local = shard_model_for_tp(model, HEADS)
            seen = {"row_parallel": 0}
            orig_out = _duo._out_linear

            def counting(be_, proj, *a):
                seen["row_parallel"] += int(_duo._row_parallel(proj))
                return orig_out(be_, proj, *a)

            _duo._out_linear = counting
            out = _run(model, local, wrap=lambda m: _with_fused_layers(m, _duo))
            assert seen["row_parallel"] == 2 * 3 * 2, seen         # 2 decode steps x 3 layers x (o_proj, down_proj)
            assert be.calls == 2 * 3 * 4
            if rank == 0:
                q.put(out.numpy())
            dist.barrier()
            return
        if mode == "patched_first":
            # the reference's harness order (eval/needle/needle_in_haystack.py:195-214, eval/LongBench/pred.py:243): the
            # DuoAttention enabler FIRST, to_device(enable_tp=True) on the already reordered model second
            from duo_attn.patch import enable_duo_attention_eval, get_full_attention_heads
            from duo_attn.utils import to_device

            enable_duo_attention_eval(model, HEADS.copy(), 4, 8)
            to_device(model, ["cpu"] * world, enable_tp=True)
            # every rank holds 2 of the 4 kv heads, retrieval heads first, dealt evenly (3 / 1 / 2 retrieval heads per layer)
            mine = [l.self_attn.full_attention_heads.tolist() for l in model.model.layers]
            assert [len(r) for r in mine] == [2, 2, 2] and all(r == sorted(r, reverse=True) for r in mine), mine
            assert [sum(r) for r in mine] == ([2.0, 0.0, 1.0] if rank == 0 else [1.0, 1.0, 1.0]), (rank, mine)     # extras to the least loaded
            # the gathered pattern is the unsharded patched model's (reordered) one
            got = torch.stack(get_full_attention_heads(model)).float().numpy()
            assert np.array_equal(got, -np.sort(-HEADS, axis=1)), got
            out = _run_tuple(model)
            if rank == 0:
                q.put(out.numpy())
            dist.barrier()
            return
        if mode != "explicit":
            # the reference's call shape (utils.py:206-227) as a drop-in: shard first, then hand the WHOLE-model pattern to
            # the enabler and the cache — they slice it to this rank's heads.  "balanced": the split knows the pattern;
            # "contiguous": it does not (rank d = the d-th block of kv heads, like the reference's tensor_parallel split)
            from duo_attn.utils import to_device

            to_device(model, ["cpu"] * world, enable_tp=True, full_attention_heads=HEADS if mode == "balanced" else None)
            if mode == "contiguous":
                assert model._duo_tp["assign"][0] == [[0, 1], [2, 3]]
            out = _run(model, HEADS)
            if rank == 0:
                q.put(out.numpy())
            dist.barrier()
            return
        local = shard_model_for_tp(model, HEADS)
        assert local.shape == (3, 2) and (np.diff(local, axis=1) <= 0).all()     # retrieval heads first
        out = _run(model, local)
        # TP-aware pattern accessors on the TUPLE-path patch (reference llama.py:601-693, TensorParallel branches):
        # every rank sees the whole model's rows in the original head order; set() scatters them back
        from duo_attn.patch import enable_duo_attention_eval, get_full_attention_heads, set_full_attention_heads

        m2 = _tiny()
        loc2 = shard_model_for_tp(m2, HEADS)
        enable_duo_attention_eval(m2, loc2.copy(), 4, 8)
        got = torch.stack(get_full_attention_heads(m2)).float().numpy()
        assert np.array_equal(got, HEADS), got
        new = torch.tensor(1.0 - HEADS, dtype=torch.float32)
        set_full_attention_heads(m2, [r for r in new])
        assert np.array_equal(torch.stack(get_full_attention_heads(m2)).float().numpy(), 1.0 - HEADS)
        # each rank kept exactly its own heads of the new rows
        for l, layer in enumerate(m2.model.layers):
            ids = m2._duo_tp["assign"][l][rank]
            assert layer.self_attn.full_attention_heads.tolist() == [float(new[l, h]) for h in ids]
        if rank == 0:
            q.put(out.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_tp2_equals_single_process():
    _paths()
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    try:
        want = _run(_tiny(), HEADS).numpy()
    finally:
        backend._set_backend_for_testing(None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # fp32 model; the two-way sums of the all-reduces change the rounding order only
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4), np.abs(got - want).max()


@pytest.mark.parametrize("mode", ["balanced", "contiguous"])
def test_to_device_enable_tp_is_a_drop_in(mode):
    """to_device(model, devices, enable_tp=True) + the whole-model pattern handed to the enabler and the KV cache."""
    _paths()
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    try:
        want = _run(_tiny(), HEADS).numpy()
    finally:
        backend._set_backend_for_testing(None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4), np.abs(got - want).max()


def test_to_device_enable_tp_after_the_enabler_like_the_reference_harnesses():
    """enable_duo_attention_eval, THEN to_device(enable_tp=True) — the order of the reference's NIAH / LongBench
    harnesses — shards the already reordered model and reproduces the single-process tuple-path logits"""
    _paths()
    from duo_attn import backend
    from duo_attn.patch import enable_duo_attention_eval
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    try:
        m = _tiny()
        enable_duo_attention_eval(m, HEADS.copy(), 4, 8)
        want = _run_tuple(m).numpy()
    finally:
        backend._set_backend_for_testing(None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "patched_first")) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4), np.abs(got - want).max()


def test_tp2_fused_decode_layers_equal_single_process():
    """the fused decode-layer form on tensor-parallel shards (row-parallel o_proj / down_proj: local product -> all-reduce
    -> residual add) == the single-process module-by-module model, fp32"""
    _paths()
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    try:
        want = _run(_tiny(), HEADS).numpy()
    finally:
        backend._set_backend_for_testing(None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "fused")) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4), np.abs(got - want).max()


def test_balanced_head_assignment():
    _paths()
    from duo_attn.tp import balanced_head_assignment

    counts = [1, 1, 2, 2, 2, 4, 2, 4, 6, 4, 5, 3, 2, 6, 5, 5, 5, 6, 3, 5, 6, 3, 3, 6, 4, 5, 3, 4, 6, 5, 8, 2]
    rng = np.random.RandomState(0)
    heads = np.zeros((32, 8))
    for l, c in enumerate(counts):
        heads[l, rng.permutation(8)[:c]] = 1.0
    for tp in (2, 4, 8):
        a = balanced_head_assignment(heads, tp)
        tot = [0] * tp
        for l in range(32):
            assert sorted(h for r in a[l] for h in r) == list(range(8))          # a partition of the layer's heads
            nr = [sum(heads[l, h] > 0.5 for h in a[l][r]) for r in range(tp)]
            assert max(nr) - min(nr) <= 1                                          # even within the layer
            for r in range(tp):
                assert len(a[l][r]) == 8 // tp
                k = [heads[l, h] for h in a[l][r]]
                assert k == sorted(k, reverse=True)                                # retrieval heads first
                tot[r] += nr[r]
        assert max(tot) - min(tot) <= 1, tot                                       # and across the whole model
        # the reference's contiguous split for comparison: rank r takes heads [r*per, (r+1)*per)
        per = 8 // tp
        ref = [sum(heads[l, r * per:(r + 1) * per].sum() for l in range(32)) for r in range(tp)]
        assert max(tot) <= max(ref)
    with pytest.raises(ValueError):
        balanced_head_assignment(heads, 3)
