"""Layer pipeline (duo_attn/pipeline.py) on CPU: world_size 2 and 3, gloo backend.

Covers the N>1 path of bench.py by construction: the even layer split, the item streaming with
batched point-to-point hand-offs (send(i) paired with recv(i+1) on the middle stages, asynchronous sends), and the sharded hot path (each rank owns the dual KV
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


def _sample(y):
    """stand-in for argmax over logits: an int64 [1, 1] token derived from the last stage's output"""
    return y.float().abs().argmax().reshape(1, 1).to(torch.int64) % 97


def _worker(rank, world, port, counts, chunks, q, subgroups=None, audit=False, _audit_log=None):
    """every run is audited (helpers.P2PAudit / check_p2p_logs: both ends of every hop issued the same way, pair
    communicators mirrored); ``audit=True`` additionally reports the hop count to the parent"""
    _setup_paths()
    if _audit_log is None:
        from helpers import P2PAudit

        with P2PAudit() as log:
            _worker(rank, world, port, counts, chunks, q, subgroups, audit, _audit_log=log)
        return
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from duo_attn import backend
        from duo_attn.pipeline import LayerPipeline
        from oracle.duo_oracle import OracleBackend

        backend._set_backend_for_testing(OracleBackend())
        Hq, Hkv, D, sink, recent = 4, 2, 128, 2, 4
        group = None
        if subgroups is not None:
            # several pipelines side by side, each on a sub-group whose ranks are NOT 0..n-1 (stage index != global
            # rank): every process creates every group (torch.distributed's rule), then works in its own
            groups = [dist.new_group(ranks=r) for r in subgroups]
            group = groups[[rank in r for r in subgroups].index(True)]
        # ragged costs -> the bottleneck-minimising split (what bench.py uses for N > 1)
        pipe = LayerPipeline(len(counts), layer_costs=[0.5 + c for c in counts], group=group)
        if subgroups is not None:
            mine = [r for r in subgroups if rank in r][0]
            assert pipe.rank == mine.index(rank) and pipe.world_size == len(mine)
            assert pipe.peer(pipe.rank) == rank and [pipe.peer(i) for i in range(len(mine))] == mine
        assert pipe.bounds[0][0] == 0 and pipe.bounds[-1][1] == len(counts)
        stage = _hot_path_stage_factory(counts, Hq, Hkv, sink, recent, sum(chunks) + 2,
                                        (pipe.first_layer, pipe.last_layer))
        g = torch.Generator().manual_seed(0)
        inputs = [torch.randn(1, S, Hq * D, generator=g).to(torch.bfloat16) for S in chunks]
        n_pre = sum(1 for S in chunks if S > 1)

        # prefill chunks: streamed (chunk c+1 enters stage 0 while chunk c is downstream)
        outs = pipe.run([(1, S, Hq * D) for S in chunks[:n_pre]],
                        lambda i, x: stage(i, inputs[i] if x is None else x, chunks[i]), device="cpu")
        # decode tokens: autoregressive — the "sampled token" of item i shapes item i+1's input
        tok = {"t": 0}

        def feedback(i, t):
            if pipe.is_last:
                return _sample(t)
            tok["t"] = int(t.item())

        def dec(i, x):
            j = n_pre + i
            return stage(j, torch.roll(inputs[j], tok["t"], dims=-1) if x is None else x, 1)

        outs += pipe.run([(1, 1, Hq * D)] * (len(chunks) - n_pre), dec, device="cpu", token_feedback=feedback)
        if _audit_log is not None:
            # what RCCL would need and gloo cannot show: both ends of every hop issued the same way (helpers.P2PAudit)
            from helpers import check_p2p_logs

            logs = [None] * world
            dist.all_gather_object(logs, list(_audit_log.calls))
            n_hops = check_p2p_logs(logs)
            if pipe.is_last and audit:
                q.put(("audit", n_hops, sorted({c[2] for lg in logs for c in lg})))
        if pipe.is_last:
            q.put([o.float().numpy() for o in outs])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelines_on_subgroups_address_global_ranks():
    """Two independent 2-stage pipelines inside one 4-rank world, on the sub-groups {1, 3} and {0, 2}: stage indices are
    group-relative, the point-to-point peers torch.distributed wants are GLOBAL ranks (LayerPipeline.peer)."""
    counts, chunks = [1, 0, 2, 1], [9, 7, 5, 1, 1, 1]
    expected = _single_process(counts, chunks)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    subgroups = [[1, 3], [0, 2]]      # (new_group orders a group's ranks ascending)
    procs = [ctx.Process(target=_worker, args=(r, 4, port, counts, chunks, q, subgroups)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]      # the last stage of each pipeline reports
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for got in results:
        assert len(got) == len(expected)
        for a, b in zip(got, expected):
            assert (a == b).all()


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
        outs, t = [], 0
        for i, S in enumerate(chunks):
            y = stage(i, torch.roll(inputs[i], t, dims=-1) if S == 1 else inputs[i], S)
            if S == 1:
                t = int(_sample(y).item())
            outs.append(y.float().numpy())
        return outs
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


@pytest.mark.parametrize("world", [3, 4])
def test_both_ends_of_every_hop_run_on_the_same_communicator(world):
    """torch's RCCL/NCCL process group (lazily initialised) runs a plain isend / irecv on the two-rank communicator of its
    pair and a call inside batch_isend_irecv on the group-wide communicator; operations on different communicators never
    match.  Until round 5 the middle stages of the prefill stream paired send(i) with recv(i+1) in a batch while the first
    and last stage issued plain calls — invisible over gloo (it matches by source and tag), a hang on three or more GPUs
    unless the group was initialised eagerly.  The audit records how every point-to-point call of every rank was issued,
    checks hop by hop that sender and receiver agree, and replays the batches on an in-order communicator to the end."""
    counts = [1, 2, 0, 1, 1, 2][: world + 2]
    chunks = [9, 7, 5, 6, 1, 1, 1]
    expected = _single_process(counts, chunks)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, chunks, q, None, True)) for r in range(world)]
    for p in procs:
        p.start()
    tag, n_hops, kinds = q.get(timeout=240)
    got = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # 7 items over world - 1 hops, + 2 token hops back (3 decode tokens: the last needs none)
    assert tag == "audit" and n_hops == 7 * (world - 1) + 2 and kinds == ["group"]     # every call batched: the group's communicator
    for a, b in zip(got, expected):
        assert (a == b).all()


def test_the_audit_catches_a_mixed_hop():
    """the audit itself: a plain send met by a batched receive is reported"""
    from helpers import check_p2p_logs

    ok = [[("send", 1, "pair")], [("recv", 0, "pair"), ("send", 2, "group")], [("recv", 1, "group")]]
    assert check_p2p_logs(ok) == 2
    bad = [[("send", 1, "pair"), ("send", 1, "pair")], [("recv", 0, "pair"), ("recv", 0, "group")]]
    with pytest.raises(AssertionError, match="never match"):
        check_p2p_logs(bad)
    with pytest.raises(AssertionError, match="1 sends, 0 receives"):
        check_p2p_logs([[("send", 1, "pair")], []])
    # counts and kinds agree, but both ranks send first on their in-order pair communicator
    crossed = [[("send", 1, "pair"), ("recv", 1, "pair")], [("send", 0, "pair"), ("recv", 0, "pair")]]
    with pytest.raises(AssertionError, match="both ranks send"):
        check_p2p_logs(crossed)
    assert check_p2p_logs([[("send", 1, "pair"), ("recv", 1, "pair")], [("recv", 0, "pair"), ("send", 0, "pair")]]) == 2
    # the group communicator runs batches in issue order: rank 0 sends to 1 first, rank 1 waits for rank 2 first, rank 2 for
    # rank 0 — nobody's head batch can retire; the same calls paired into batches can
    ring = [[("send", 1, "group", 1), ("recv", 2, "group", 2)], [("send", 2, "group", 1), ("recv", 0, "group", 2)],
            [("send", 0, "group", 1), ("recv", 1, "group", 2)]]
    with pytest.raises(AssertionError, match="cannot make progress"):
        check_p2p_logs(ring)
    assert check_p2p_logs([[(d, p, k, 1) for d, p, k, _ in lg] for lg in ring]) == 3


def test_balanced_layer_split():
    from duo_attn.utils import balanced_layer_split, even_layer_split

    counts = [1, 1, 2, 2, 2, 4, 2, 4, 6, 4, 5, 3, 2, 6, 5, 5, 5, 6, 3, 5, 6, 3, 3, 6, 4, 5, 3, 4, 6, 5, 8, 2]
    costs = [0.5 + c for c in counts]
    for P in (1, 2, 4, 8, 32):
        b = balanced_layer_split(costs, P)
        assert len(b) == P and b[0][0] == 0 and b[-1][1] == 32
        assert all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(e > s for s, e in b)
        worst = max(sum(costs[s:e]) for s, e in b)
        even = max(sum(costs[s:e]) for s, e in even_layer_split(32, P))
        assert worst <= even
        # exhaustive optimum for P = 2
        if P == 2:
            assert worst == min(max(sum(costs[:c]), sum(costs[c:])) for c in range(1, 32))
    assert balanced_layer_split([1, 1, 1, 1], 2) == [(0, 2), (2, 4)]
    with pytest.raises(ValueError):
        balanced_layer_split([1, 1], 3)


def test_even_layer_split():
    from duo_attn.utils import even_layer_split

    assert even_layer_split(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert even_layer_split(32, 1) == [(0, 32)]
    b = even_layer_split(5, 3)
    assert b[0][0] == 0 and b[-1][1] == 5 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all(e > s for s, e in b)
