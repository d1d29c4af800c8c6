"""Two layer blocks per rank (duo_attn/pipeline.py: InterleavedLayerPipeline) on CPU: world sizes 2, 3 and 4, gloo backend.

VERDICT r5 item 7: 32 ragged layers cut into 8 contiguous pieces leave the idlest pipeline stage 0.61 busy; with two blocks per
rank (rank r owns blocks r and P + r, an item goes round the ranks twice) the cuts can pair a heavy block with a light one.
Code and CPU tests only — there is no multi-GPU box: the sharded hot path equals a single-process run (chunked prefill
streamed in groups of P items, then autoregressive decode with the token fed back), every point-to-point call is audited for
the communicator RCCL would run it on and the batches are REPLAYED on an in-order communicator in both initialisation modes
(helpers.P2PAudit / check_p2p_logs: the ring — rank P-1 hands pass 0's output back to rank 0 — is where a receive posted too
early would stop everything).  Reference: the contiguous layer sharding of duo_attn/utils.py:251-271, cut finer."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_pipeline_gloo import _free_port, _hot_path_stage_factory, _sample, _setup_paths, _single_process  # noqa: E402


def _worker(rank, world, port, counts, chunks, q, gs=None, _audit_log=None):
    _setup_paths()
    if _audit_log is None:
        from helpers import P2PAudit

        with P2PAudit() as log:
            _worker(rank, world, port, counts, chunks, q, gs, _audit_log=log)
        return
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from duo_attn import backend
        from duo_attn.pipeline import InterleavedLayerPipeline
        from helpers import check_p2p_logs
        from oracle.duo_oracle import OracleBackend

        backend._set_backend_for_testing(OracleBackend())
        Hq, Hkv, D, sink, recent = 4, 2, 128, 2, 4
        pipe = InterleavedLayerPipeline(len(counts), layer_costs=[0.5 + c for c in counts], group_size=gs)
        b = pipe.bounds
        assert len(b) == 2 * world and b[0][0] == 0 and b[-1][1] == len(counts) and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert pipe.blocks == (b[rank], b[world + rank])
        # one dual KV cache per owned block; the toy layer of test_pipeline_gloo.py
        stages = [_hot_path_stage_factory(counts, Hq, Hkv, sink, recent, sum(chunks) + 2, blk) for blk in pipe.blocks]
        g = torch.Generator().manual_seed(0)
        inputs = [torch.randn(1, S, Hq * D, generator=g).to(torch.bfloat16) for S in chunks]
        n_pre = sum(1 for S in chunks if S > 1)
        order = []

        def pre(i, x, ps):
            order.append((i, ps))
            return stages[ps](i, inputs[i] if x is None else x, chunks[i])

        outs = pipe.run([(1, S, Hq * D) for S in chunks[:n_pre]], pre, device="cpu")
        # the schedule: groups of `world` items, pass 0 of a group then its pass 1 — every layer block sees its items in order
        assert pipe.group_size == (gs or pipe.group_size) and pipe.group_size >= 1 and order == pipe.units(n_pre, pipe.group_size)
        for ps in (0, 1):
            assert [i for i, p in order if p == ps] == list(range(n_pre))
        tok = {"t": 0}

        def feedback(i, t):
            if pipe.is_last:
                return _sample(t)
            tok["t"] = int(t.item())

        def dec(i, x, ps):
            j = n_pre + i
            return stages[ps](j, torch.roll(inputs[j], tok["t"], dims=-1) if x is None else x, 1)

        outs += pipe.run([(1, 1, Hq * D)] * (len(chunks) - n_pre), dec, device="cpu", token_feedback=feedback)
        logs = [None] * world
        dist.all_gather_object(logs, list(_audit_log.calls))
        n_hops = check_p2p_logs(logs)
        if pipe.is_last:
            q.put(("audit", n_hops, sorted({c[2] for lg in logs for c in lg})))
            q.put([o.float().numpy() for o in outs])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,counts,chunks,gs", [
    (2, [1, 0, 2, 1], [9, 7, 5, 1, 1, 1], 2),                        # three prefill items in groups of two: a last group of ONE
    (2, [1, 0, 2, 1, 2, 0], [6, 6, 6, 6, 5, 1, 1], 4),               # groups twice as long as the ring, a partial last one
    (2, [1, 0, 2, 1, 2, 0], [6, 6, 6, 6, 5, 1, 1], None),            # the planner's own choice
    (3, [1, 2, 0, 1, 1, 0, 2], [9, 7, 5, 4, 1, 1, 1], 3),
    (3, [1, 2, 0, 1, 1, 0, 2], [4, 4, 4, 4, 4, 4, 4, 3, 1, 1], 6),   # eight items, groups of six: rank 0 holds several inputs in flight
    (4, [1, 2, 0, 1, 1, 0, 2, 1, 1], [5, 5, 5, 5, 5, 5, 5, 5, 3, 1, 1], 4),    # nine items on four ranks: two full groups + one item
    (4, [1, 2, 0, 1, 1, 0, 2, 1, 1], [5, 5, 5, 5, 5, 5, 5, 5, 3, 1, 1], 8),
])
def test_two_blocks_per_rank_equal_a_single_process(world, counts, chunks, gs):
    expected = _single_process(counts, chunks)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, chunks, q, gs)) for r in range(world)]
    for p in procs:
        p.start()
    audit = q.get(timeout=240)
    got = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_pre, n_dec = sum(1 for S in chunks if S > 1), sum(1 for S in chunks if S == 1)
    # hops: every unit but the last rank's pass 1 is sent on — (2 world - 1) sends per item — plus one token per decode step but the last
    assert audit[0] == "audit" and audit[1] == (n_pre + n_dec) * (2 * world - 1) + (n_dec - 1) and audit[2] == ["group"]
    assert len(got) == len(expected)
    for a, b in zip(got, expected):
        assert (a == b).all()


def test_the_split_minimises_the_simulated_makespan():
    """the bench workload's own per-layer prefill cost (Llama-3-8B pattern at 50 %, 131 072 tokens, chunk 16 384): every layer
    owned exactly once; the simulated stream of the chosen plan is never slower than one block per rank, 4 % faster on two
    ranks and 7 % on four (on eight ranks 32 layers are too few to cut finer: 16 blocks of two layers — about even)"""
    _setup_paths()
    import bench
    from duo_attn.pipeline import interleaved_layer_split, simulate_interleaved
    from duo_attn.utils import balanced_layer_split

    counts = bench.LLAMA3_8B_FULL_KV_HEADS
    pf = bench.prefill_flops(counts, 131072, 16384)
    cost = [sum(pf[c][li] for c in range(len(pf))) for li in range(len(counts))]
    tot = sum(cost)
    for P, gain in ((2, 1.04), (4, 1.07), (8, 1.0)):
        b, gs = interleaved_layer_split(cost, P)
        assert gs in (P, 2 * P)
        assert len(b) == 2 * P and b[0][0] == 0 and b[-1][1] == len(cost) and all(x[1] == y[0] and x[0] < x[1] for x, y in zip(b, b[1:]))
        n = 8 * P
        two = simulate_interleaved([sum(cost[lo:hi]) / tot for lo, hi in b], P, gs, n)
        one = [sum(cost[lo:hi]) / tot for lo, hi in balanced_layer_split(cost, P)]
        fin = [[0.0] * n for _ in range(P)]
        for st in range(P):
            for i in range(n):
                fin[st][i] = max(fin[st][i - 1] if i else 0.0, fin[st - 1][i] if st else 0.0) + one[st]
        assert two <= fin[-1][-1] / gain + 1e-9, (P, two, fin[-1][-1])
        assert two >= n / P - 1e-9                  # never below the work-conserving bound
    with pytest.raises(ValueError):
        interleaved_layer_split([1.0] * 5, 3)
