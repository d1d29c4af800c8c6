"""Layer pipeline (duo_attn/pipeline.py) on CPU: world_size 2 and 3, gloo backend.

Covers the N>1 path of bench.py by construction: the even layer split, the item streaming with
posted-ahead receives / asynchronous sends, and the sharded hot path (each rank owns the dual KV
pools of its layers) against a single-process run.  The oracle is the device backend here.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "duo-attention_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _hot_path_stage_factory(counts, Hq, Hkv, sink, recent, max_size, layer_range):
    from helpers import ShapeModel, heads_from_counts
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    D = 128
    l0, l1 = layer_range
    local = counts[l0:l1]
    cache = DuoAttentionStaticKVCache(ShapeModel(len(local), Hq, Hkv, D), heads_from_counts(local, Hkv), 1,
                                      max_size, sink, recent)
    state = {"pos": 0}

    def stage(i, x, S):
        # x: hidden [1, S, Hq*D]; a toy "layer": q = x, k/v = column slices, residual add of the attention
        for li in range(len(local)):
            q = x.clone().view(1, S, Hq, D)
            k = x[..., : Hkv * D].clone().view(1, S, Hkv, D)
            v = x[..., -Hkv * D:].clone().view(1, S, Hkv, D)
            a = duo_static_attention_core(q, k, v, cache, li, state["pos"], 1.0, 10000.0)
            x = (x.float() + 0.5 * a.reshape(1, S, Hq * D).float()).to(torch.bfloat16)
        state["pos"] += S
        return x

    return stage


def _worker(rank, world, port, counts, chunks, q):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from duo_attn import backend
        from duo_attn.pipeline import LayerPipeline
        from oracle.duo_oracle import OracleBackend

        backend._set_backend_for_testing(OracleBackend())
        Hq, Hkv, D, sink, recent = 4, 2, 128, 2, 4
        pipe = LayerPipeline(len(counts))
        stage = _hot_path_stage_factory(counts, Hq, Hkv, sink, recent, sum(chunks) + 2,
                                        (pipe.first_layer, pipe.last_layer))
        g = torch.Generator().manual_seed(0)
        inputs = [torch.randn(1, S, Hq * D, generator=g).to(torch.bfloat16) for S in chunks]

        def fn(i, x):
            return stage(i, inputs[i] if x is None else x, chunks[i])

        outs = pipe.run([(1, S, Hq * D) for S in chunks], fn, device="cpu")
        if pipe.is_last:
            q.put([o.float().numpy() for o in outs])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _single_process(counts, chunks):
    _setup_paths()
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend())
    try:
        Hq, Hkv, D, sink, recent = 4, 2, 128, 2, 4
        stage = _hot_path_stage_factory(counts, Hq, Hkv, sink, recent, sum(chunks) + 2, (0, len(counts)))
        g = torch.Generator().manual_seed(0)
        inputs = [torch.randn(1, S, Hq * D, generator=g).to(torch.bfloat16) for S in chunks]
        return [stage(i, inputs[i], S).float().numpy() for i, S in enumerate(chunks)]
    finally:
        backend._set_backend_for_testing(None)


@pytest.mark.parametrize("world,counts", [(2, [1, 0, 2, 1]), (3, [1, 2, 0, 1, 1])])
def test_sharded_hot_path_equals_single_process(world, counts):
    chunks = [9, 7, 5, 1, 1, 1]       # three prefill chunks then three decode tokens
    expected = _single_process(counts, chunks)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == len(expected)
    for a, b in zip(got, expected):
        assert (a == b).all()


def test_even_layer_split():
    from duo_attn.utils import even_layer_split

    assert even_layer_split(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert even_layer_split(32, 1) == [(0, 32)]
    b = even_layer_split(5, 3)
    assert b[0][0] == 0 and b[-1][1] == 5 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all(e > s for s, e in b)
