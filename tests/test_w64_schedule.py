"""Legality of the generated bulk-tile schedule of the 4-wave prefill kernel (tools/gen_w64_bulk.py): the hazards the
asm statements rely on are properties of the placement table, so they are checked on the table (CPU, no GPU)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_w64_bulk", os.path.join(ROOT, "tools", "gen_w64_bulk.py"))
    mod = importlib.util.module_from_spec(spec)
    env = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("W64_GEN_")}
    try:
        spec.loader.exec_module(mod)
    finally:
        os.environ.update(env)
    return mod


def _ops(gaps):
    """(gap, position in the gap, op) for every op of a variant"""
    return [(g, i, op) for g, ops in enumerate(gaps) for i, op in enumerate(ops)]


def _writes(op, pattern):
    return [c for _, _, c, acc in op.operands if acc in ("w", "rw") and re.fullmatch(pattern, c)]


def _reads(op, pattern):
    return [c for _, _, c, acc in op.operands if acc in ("r", "rw") and re.fullmatch(pattern, c)]


def test_every_instruction_of_a_tile_is_placed_exactly_once():
    gen = _gen()
    steady = gen.build_gaps("STEADY")
    texts = [op.text.split()[0] for _, _, op in _ops(steady)]
    assert texts.count("@MFMA@") == 64
    assert texts.count("v_fma_f32") == 64 and texts.count("v_exp_f32") == 64 and texts.count("v_add_f32") == 64
    assert texts.count("@CVT@") == 32
    assert texts.count("ds_read_b64_tr_b16") == 32 and texts.count("ds_read_b128") == 16
    assert texts.count("global_load_lds_dwordx4") == 8
    # every packed P word, every V^T half-fragment and every K fragment is written exactly once per tile
    for pat, n in ((r"pk[AB]\[\d+\]", 32), (r"v(lo|hi)\[\d+\]", 32)):
        w = [c for _, _, op in _ops(steady) for c in _writes(op, pat)]
        assert len(w) == n and len(set(w)) == n
    kfr = sorted(int(m.group(1)) for _, _, op in _ops(steady) for m in [re.match(r"ds_read_b128 a\[(\d+):", op.text)] if m)
    assert kfr == list(range(192, 256, 4))
    # FIRST drops block B(t-1)'s work and the ph2 MFMAs, DRAIN is exactly that work
    first, drain = gen.build_gaps("FIRST"), gen.build_gaps("DRAIN")
    n = lambda gaps: sum(op.n for _, _, op in _ops(gaps) if not op.text.startswith("@MFMA@"))
    m = lambda gaps: sum(1 for _, _, op in _ops(gaps) if op.text.startswith("@MFMA@"))
    assert m(first) == 48 and m(drain) == 16
    assert n(first) + n(drain) == n(steady)


def test_dependencies_and_hazards_of_the_steady_state():
    gen = _gen()
    gaps = gen.build_gaps("STEADY")
    ops = _ops(gaps)
    fin = {x: next(g for g, _, op in ops if "v_cmp_gt_f32" in op.text and f"mk{x}" in str(op.operands)) for x in "AB"}
    for g, i, op in ops:
        head = op.text.split()[0]
        # scores are read only after their last MFMA is three MFMAs behind (S_A: MFMAs 14/15 of ph1, S_B: of ph3); block B's
        # scores of tile t-1 are read in gaps 0..16 of the next tile, before ph3 overwrites them
        for c in _reads(op, r"s[ab]\[\d\]\[\d+\]"):
            half = int(c[3])
            if c[1] == "a":
                assert g >= 17 + half, (g, c)
            else:
                assert g >= 49 + half or g <= 26, (g, c)
        # exponentiation uses the max decided in the FIN statement of its block (same tile, or the previous one for B)
        if head == "v_fma_f32":
            x = "A" if _reads(op, r"sa\[.*") else "B"
            assert g > fin[x] or (x == "B" and g <= 26), (g, x)
        # V^T fragment f is free once ph2 MFMA 16+f has read it, and must land before the barrier in front of ph4
        for c in _writes(op, r"v(lo|hi)\[\d+\]"):
            f = int(re.search(r"\[(\d+)\]", c).group(1))
            assert 17 + f <= g <= 41, (g, c)
        if head == "ds_read_b128":
            assert g >= 48            # K(t+1) fragments: a[192:255] is read by ph1 and ph3 of tile t
        if "global_load_lds" in op.text:
            assert g < 48             # counted vmcnt(8) at the barrier: the 8 pieces of tile t+2 are the youngest loads
    # a packed P word is written at least two gaps before the MFMA that consumes its k-slot
    for x, first_mfma in (("A", 48), ("B", 16)):
        for g, i, op in ops:
            for c in _writes(op, rf"pk{x}\[\d+\]"):
                j = int(re.search(r"\[(\d+)\]", c).group(1))
                need = first_mfma + 4 * (j >> 2)
                gg = g if x == "A" or g > 26 else g + 64      # block B's late words are written in the next tile's ph1 / ph2
                if x == "B":
                    need += 64
                assert gg <= need - 2, (x, j, g)
    # the rule of the table: a statement never reads or writes a register the statement before it wrote (the compiler
    # assumes a 16-bit destination select in every asm statement and pads an s_nop otherwise) — row sums alternate
    # between two accumulators, the row max between two pairs of chains, slice consumers sit two gaps behind
    def regs(stmt, accs):
        return {c for op in stmt for _, cls, c, acc in op.operands if cls == "v" and acc in accs and not c.startswith("W64_")}
    for g in range(64):
        nxt = gaps[(g + 1) % 64]
        shared = regs(gaps[g], ("w", "rw")) & regs(nxt, ("r", "w", "rw"))
        assert not shared, (g, shared)
    # no instruction reads a transcendental's result in the very next slot of the same statement
    for g, stmt in enumerate(gaps):
        for a, b in zip(stmt, stmt[1:]):
            if a.text.startswith("v_exp_f32"):
                tgt = _writes(a, r"e[AB]\[\d\]\[\d\]")
                assert not set(tgt) & set(_reads(b, r"e[AB]\[\d\]\[\d\]")), (g, a.text, b.text)
    # an LDS-DMA load has its M0 write in the same statement with at least one instruction in between
    for g, stmt in enumerate(gaps):
        idx = [i for i, op in enumerate(stmt) if op.text.startswith("s_mov_b32 m0")]
        ld = [i for i, op in enumerate(stmt) if op.text.startswith("global_load_lds")]
        assert len(idx) == len(ld)
        for i, j in zip(idx, ld):
            assert j - i >= 2, g


def test_gap_loads_are_balanced():
    gen = _gen()
    loads = [sum(op.n for op in stmt if not op.text.startswith("@MFMA@")) for stmt in gen.build_gaps("STEADY")]
    assert sum(loads) == 330
    assert max(loads) <= 8 and sorted(loads)[4] >= 4      # the two FIN gaps carry 7-8, the gap before block A's FIN 1


def test_xmap_is_a_bijection():
    """XCD-aware block order of the retrieval class (duo_prefill_w64_kernel.inc + the launcher in duo_prefill.hip), restated:
    every (q tile, kv head, q-head-in-group) exactly once, the padded blocks of the last period leave, and an XCD
    (block id % 8) sees at most two kv heads."""
    def blocks(nf, G, nqt):
        row_items = nf * G
        rows = 1
        while (rows * row_items) % 8:
            rows *= 2
        periods = (nqt + rows - 1) // rows
        q = rows * row_items // 8
        seen, per_xcd = set(), {}
        for b in range(periods * rows * row_items):
            x, r = b & 7, b >> 3
            per, j = divmod(r, q)
            w = x * q + j
            kvh, e = divmod(w, rows * G)
            row, g = divmod(e, G)
            rank = per * rows + row
            if rank >= nqt:
                continue
            item = (nqt - 1 - rank, kvh, g)
            assert 0 <= kvh < nf and item not in seen
            seen.add(item)
            per_xcd.setdefault(x, set()).add(kvh)
        assert len(seen) == nf * G * nqt
        return max(len(v) for v in per_xcd.values())

    for nf in range(1, 9):
        for G in (1, 2, 3, 4, 6, 8):
            for nqt in (1, 2, 3, 7, 64, 125):
                heads = blocks(nf, G, nqt)
                if G == 4 and nqt == 64:
                    assert heads <= 2


def test_w64_kernels_own_their_accumulator_file():
    """tools/debug/audit_w64.sh on the current sources: the 4-wave prefill kernels address the accumulator half of the
    register file by literal register numbers inside asm statements, so the compiler must not touch it — no
    compiler-generated instruction may name an AGPR, 256 AGPRs / scratch 0 must be what the kernel descriptor says"""
    import os
    import shutil
    import subprocess

    import pytest

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not present")
    r = subprocess.run(["bash", os.path.join(root, "tools", "debug", "audit_w64.sh"), root], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("duo_prefill_w64")]
    assert len(lines) == 2 and all("'NumAgprs': '256'" in l and "'ScratchSize': '0'" in l and l.endswith("AGPRs: 0") for l in lines), lines
