"""CPU: the prefill launcher's plan (csrc/duo_prefill.hip) — key-range pieces per head class chosen by replaying the launch on
256 CUs, and the block -> work mapping both prefill kernels and the merge kernel share (prefill_map_block,
csrc/duo_prefill_common.h).  Host code only: ``duo_debug_prefill_plan`` computes the plan and the mapping without a GPU.

What the flash_attn_func calls this launch replaces compute is fixed (reference llama.py:366-372, :392-421); HOW the work is cut
over workgroups must never change it: every (class, q tile, q head, piece) exactly once, every partial slot exactly once."""
import numpy as np
import pytest

from duo_attn import _hip

QBLK, KV = 256, 64


def _check_cover(nkv0, nkv1, group, S, r, xmap):
    m = r["map"]
    assert len(m) == r["blocks"]
    real = m[m[:, 1] >= 0]
    ks = (r["k0"], r["k1"])
    nq = (S + QBLK - 1) // QBLK
    want = set()
    for c, nkv in ((0, nkv0), (1, nkv1)):
        for t in range(nq):
            for h in range(nkv):
                for g in range(group):
                    for s in range(ks[c]):
                        want.add((c, t, h, g, s))
    got = [tuple(int(v) for v in row[:5]) for row in real]
    assert len(got) == len(set(got)) == len(want) and set(got) == want
    # class 0 sits in front, padded to a multiple of 8 blocks (class 1's block ids keep b % 8 = XCD)
    assert (m[:r["blocks0"], 0] == 0).all() and (m[r["blocks0"]:, 0] == 1).all()
    assert nkv1 == 0 or nkv0 == 0 or r["blocks0"] % 8 == 0
    # partial slots: one per piece of a split class, dense, class 0's first
    parts = [int(row[5]) for row in real if ks[int(row[0])] > 1]
    assert sorted(parts) == list(range(r["partials"]))
    n0 = nkv0 * group * nq * ks[0] if ks[0] > 1 else 0
    for row in real:
        if ks[int(row[0])] > 1:
            assert (int(row[5]) < n0) == (int(row[0]) == 0)
    # the pieces of one item are adjacent partial slots in piece order (the merge kernel walks them so)
    for row in real:
        c, t, h, g, s, p = (int(v) for v in row)
        if ks[c] > 1:
            rank = nq - 1 - t
            nkv = nkv0 if c == 0 else nkv1
            base = (0 if c == 0 else n0) + ((rank * nkv + h) * group + g) * ks[c]
            assert p == base + s


@pytest.mark.parametrize("xmap", [0, 1, 3])
def test_every_piece_of_every_item_is_mapped_exactly_once(xmap):
    rng = np.random.default_rng(5 + xmap)
    for _ in range(120):
        nkv0, nkv1 = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        if nkv0 + nkv1 == 0:
            nkv0 = 1
        group = int(rng.choice([1, 2, 3, 4, 7, 8]))
        S = int(rng.choice([1, 100, 256, 257, 1000, 2048, 4096, 5000]))
        lenB = S + int(rng.choice([0, 0, 64, 1000, 12288]))
        lenA0 = int(rng.choice([0, 1, 63, 64, 5000, 114688]))
        f0, f1 = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 16])), int(rng.choice([0, 0, 1, 2, 4]))
        force = f0 | (f1 << 8)
        r = _hip.prefill_plan(nkv0, nkv1, group, S, lenA0, lenB, min(lenA0, 384), lenB, max_parts=int(rng.choice([0, 64, 2048])),
                              xmap=xmap, force=force, with_blocks=True)
        _check_cover(nkv0, nkv1, group, S, r, xmap)
        # forced counts are clamped to what is legal: never more pieces than tiles, never more partials than the workspace
        nA = (lenA0 + KV - 1) // KV
        min_tiles0 = nA + (min(QBLK - 1, S - 1) + lenB - S) // KV + 1
        assert r["k0"] <= max(1, min_tiles0) and r["k0"] <= 16 and r["k1"] <= 16
        if f0 and nkv0:
            assert r["k0"] <= f0


def test_partials_never_exceed_the_workspace():
    for cap in (0, 10, 100, 500, 2048):
        r = _hip.prefill_plan(3, 5, 4, 2048, 114688, 16384, 384, 16384, max_parts=cap)
        assert r["partials"] <= cap
        if cap == 0:
            assert (r["k0"], r["k1"]) == (1, 1)


def test_xcd_aware_order_keeps_a_stream_on_few_xcds():
    """one K/V stream = one (kv head, key-range piece): with the XCD-aware order the workgroups an XCD (block id % 8) runs at the
    same time — 32 consecutive ones of its share — touch at most three streams, for every head / piece count"""
    for nkv0, k0 in ((3, 5), (5, 3), (6, 4), (1, 8), (7, 1), (4, 2)):
        r = _hip.prefill_plan(nkv0, 8 - nkv0, 4, 2048, 114688, 16384, 384, 16384, xmap=3, force=k0 | (1 << 8), with_blocks=True)
        assert r["k0"] == k0
        m = r["map"][:r["blocks0"]]
        for x in range(8):
            mine = m[x::8]
            mine = mine[mine[:, 1] >= 0]
            for i in range(0, len(mine), 32):
                streams = {(int(a), int(b)) for a, b in mine[i:i + 32][:, [2, 4]]}
                assert len(streams) <= 3, (nkv0, k0, x, streams)


def test_the_planner_splits_row_blocks_and_leaves_full_launches_alone():
    # a whole 16384-row chunk of a layer with 8 retrieval heads: 2048 equal workgroups = 8 full rounds: nothing to gain
    r = _hip.prefill_plan(8, 0, 4, 16384, 114688, 16384, 0, 0)
    assert (r["k0"], r["k1"]) == (1, 1)
    # 2048-row blocks of the layer pipeline (bench.py --row-block 2048) late in a 128K prompt: every ragged layer is split,
    # and the replayed launch is far shorter than the unsplit one
    for nf in (1, 2, 3, 4, 5, 6):
        r = _hip.prefill_plan(nf, 8 - nf, 4, 2048, 114688, 16384, 384, 16384)
        assert r["k0"] > 1 and r["est_us"] < 0.9 * r["est_unsplit_us"], (nf, r)
    # the estimate of the chosen plan is never above the unsplit one
    rng = np.random.default_rng(1)
    for _ in range(40):
        nf = int(rng.integers(0, 9))
        S = int(rng.choice([1024, 2048, 4096, 16384]))
        past = int(rng.choice([0, 16384, 65536, 114688]))
        r = _hip.prefill_plan(nf, 8 - nf, 4, S, past, 16384, min(past, 384), 16384)
        assert r["est_us"] <= r["est_unsplit_us"] + 1e-6
