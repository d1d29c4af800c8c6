"""bench.py --block-streams N (one GPU, measurement): consecutive row blocks of the prefill issued on alternating HIP streams, ordered
layer by layer with events.  Every launch's output and the pools afterwards must equal the one-stream run bit for bit — same
kernels, same launch plans, only the overlap differs."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod_streams", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [512, 1024])
def test_row_blocks_on_alternating_streams_equal_one_stream(rows):
    B = _bench()
    dev = torch.device("cuda", 0)
    counts, ctx, chunk = [4, 1, 8, 0, 3, 5, 2], 3 * 2048, 2048
    runs = []
    for n in (1, 2, 3):
        hp = B.HotPath(counts, (0, len(counts)), ctx, chunk, dev)
        hp.set_row_blocks(rows)
        hp.set_block_streams(n)
        hp.keep_outputs = []
        hp.cache.clear()
        for i in range(len(hp.blocks)):
            hp.prefill_block_stage(i, None)
        torch.cuda.synchronize()
        c = hp.cache
        pools = []
        for li in range(len(counts)):
            pools += [c.full_key_states_list[li], c.full_value_states_list[li], *c.get_streaming_kv(li)]
        pools = [p for p in pools if p is not None]
        runs.append(([o.clone() for o in hp.keep_outputs], [p.clone() for p in pools], list(c.kv_seq_len_list)))
        hp.free()
    ref = runs[0]
    assert len(ref[0]) == len(counts) * (ctx // rows)
    for n, run in zip((2, 3), runs[1:]):
        assert run[2] == ref[2]
        for k, (a, b) in enumerate(zip(run[0], ref[0])):
            assert torch.equal(a, b), f"{n} streams: output of launch {k} differs from the one-stream run"
        for k, (a, b) in enumerate(zip(run[1], ref[1])):
            assert torch.equal(a, b), f"{n} streams: pool tensor {k} differs from the one-stream run"


@pytest.mark.gpu
def test_whole_chunks_on_alternating_streams_equal_the_serial_run():
    """the same with whole chunks as the items (chunk i + 1 at layer l only needs chunk i's rows of layer l): two and three
    streams against the same code path serialised on one stream"""
    B = _bench()
    dev = torch.device("cuda", 0)
    counts, ctx, chunk = [4, 1, 8, 0, 3, 5, 2], 5 * 1024, 1024
    for n in (2, 3):
        runs = []
        for serial in (True, False):
            hp = B.HotPath(counts, (0, len(counts)), ctx, chunk, dev)
            hp.set_block_streams(n, serial=serial)
            hp.keep_outputs = []
            hp.cache.clear()
            for i in range(len(hp.chunks)):
                hp.prefill_stage(i, None)
            torch.cuda.synchronize()
            c = hp.cache
            pools = []
            for li in range(len(counts)):
                pools += [c.full_key_states_list[li], c.full_value_states_list[li], *c.get_streaming_kv(li)]
            runs.append(([o.clone() for o in hp.keep_outputs], [p.clone() for p in pools if p is not None], list(c.kv_seq_len_list)))
            hp.free()
        (o0, p0, l0), (o1, p1, l1) = runs
        assert l0 == l1 and len(o0) == len(o1) == len(counts) * (ctx // chunk)
        for k, (a, b) in enumerate(zip(o1, o0)):
            assert torch.equal(a, b), f"{n} streams: output of launch {k} differs from the serial run"
        for k, (a, b) in enumerate(zip(p1, p0)):
            assert torch.equal(a, b), f"{n} streams: pool tensor {k} differs from the serial run"
