#!/usr/bin/env python3
"""Generates duo-attention_amd/csrc/duo_prefill_w64_bulk.inc: the instruction schedule of ONE bulk tile of the
4-wave x 64-row prefill kernel (duo_prefill_w64.h), as a sequence of asm statements — one per MFMA "gap".

Why a generator: with one wave per SIMD every instruction next to the 64 MFMAs of a tile is a serial issue slot that
delays the next MFMA by 2.5-5 cycles (an LDS-DMA load by ~38, two loads in one gap by much more), see
profiles/r2_prefill_w64.md.  A tile carries 330 such instructions (2 x 112 exponentiation-slice instructions, 2 x 21
row-max instructions, 48 LDS fragment reads, 8 LDS-DMA loads + 8 M0 writes): 5.2 per gap ONLY IF they are spread
evenly over all 64 gaps.  Block B's softmax therefore lags block A's by half a tile (the skewed order below), and the
placement is a table in this file instead of hand-expanded macros; the W64_GEN_* environment switches rebuild the
alternatives that were measured against it (tools/debug/build_variant.sh + w64_timing_variants.sh).

Tile t, 64 gaps (gap g = the instructions issued behind MFMA g):
    ph1  g  0..15   S_A(t)  = K(t) . Q_A^T        | block B(t-1) slices (rest), LDS-DMA of tile t+2: K pieces g 0..3, V pieces g 14..17
    ph2  g 16..31   O_B    += V^T(t-1) . P_B(t-1) | row max A(t) g 17..20 + g 22, block A(t) slices from g 23, V^T(t) reads (one per gap)
    ph3  g 32..47   S_B(t)  = K(t) . Q_B^T        | V^T(t) read burst (five per gap g 32..35; spreading them measured 0.6 % slower), block A(t) slices
         -- lgkmcnt(0), vmcnt(8), ONE s_barrier --
    ph4  g 48..63   O_A    += V^T(t) . P_A(t)     | K(t+1) fragment reads (one per gap), row max B(t) g 49..52 + g 54, block B(t) slices from g 55

Variants: STEADY (above), FIRST (first tile of a bulk run: no block B(t-1) work, no ph2 MFMAs) and DRAIN (after
the last tile of a run: the pending block-B slices and the 16 ph2 MFMAs only).

Usage: python tools/gen_w64_bulk.py            (rewrites the .inc next to duo_prefill_w64.h)
"""
import os
import sys


class Op:
    def __init__(self, text, operands, n=1, clobbers=()):
        self.text = text            # "{key}" placeholders; several lines allowed
        self.operands = operands    # (key, cls, cexpr, access)   cls: v s n sout; access: r w rw
        self.n = n
        self.clobbers = tuple(clobbers)


AB = "AB"
ab = "ab"


def mfma_qk(x, i):
    acc = f"s{ab[x]}[{i & 1}]"
    k0 = 192 + 4 * i
    q0 = 128 + 32 * x + 4 * (i >> 1)
    tail = "0" if i < 2 else "{acc}"
    return Op(f"@MFMA@ {{acc}}, a[{k0}:{k0 + 3}], a[{q0}:{q0 + 3}], {tail}",
              [("acc", "v", acc, "w" if i < 2 else "rw")])


def mfma_pv(x, i):
    o0 = 64 * x + 16 * (i & 3)
    return Op(f"@MFMA@ a[{o0}:{o0 + 15}], {{vf}}, {{pf}}, a[{o0}:{o0 + 15}]",
              [("vf", "v", f"W64_VF({i})", "r"), ("pf", "v", f"W64_PF{AB[x]}({i >> 2})", "r")])


def S(x, k, h):
    return f"s{ab[x]}[{k >> 3}][{2 * (k & 7) + h}]"


NT = 4   # slices in flight per block (asserted by the scheduler): temporaries e?[k % NT][h]


def E(x, k, h):
    return f"e{AB[x]}[{k % NT}][{h}]"


def fma(x, k, h):
    return Op("v_fma_f32 {e}, {s}, {c}, {nm}",
              [("e", "v", E(x, k, h), "w"), ("s", "v", S(x, k, h), "r"), ("c", "s", "c", "r"),
               ("nm", "v", f"nmv[{x}]", "r")])


def exp(x, k, h):
    return Op("v_exp_f32 {e}, {e}", [("e", "v", E(x, k, h), "rw")])


def add(x, k, h, parity):
    # two row-sum accumulators per block, alternating by gap: no statement reads what the previous one wrote
    acc = f"lsum[{x}]" if parity == 0 else f"lodd[{x}]"
    return Op("v_add_f32 {l}, {l}, {e}", [("l", "v", acc, "rw"), ("e", "v", E(x, k, h), "r")])


ROWSUM = os.environ.get("W64_GEN_ROWSUM", "add")     # add: one fp32 v_add per score;  dot2: one packed dot per P word (sums the ROUNDED P)


def dot2(x, k, parity):
    # l += p_lo * 1 + p_hi * 1 on the packed word the PV MFMA reads: one instruction for two scores
    acc = f"lsum[{x}]" if parity == 0 else f"lodd[{x}]"
    return Op("@DOT2@ {l}, {one}, {pk}", [("l", "v", acc, "rw"), ("one", "s", "w64_ones", "r"), ("pk", "v", f"pk{AB[x]}[{k}]", "r")])


def cvt(x, k):
    return Op("@CVT@ {pk}, {e0}, {e1}",
              [("pk", "v", f"pk{AB[x]}[{k}]", "w"), ("e0", "v", E(x, k, 0), "r"), ("e1", "v", E(x, k, 1), "r")])


def e8(x, b, r0):
    return [(f"e{j}", "v", f"s{ab[x]}[{b}][{r0 + j}]", "r") for j in range(8)]


def max_init(x, b, r0, c0, c1):
    return Op("v_max3_f32 {h0}, {e0}, {e1}, {e2}\n"
              "v_max3_f32 {h1}, {e3}, {e4}, {e5}\n"
              "v_max3_f32 {h0}, {h0}, {e6}, {e7}",
              [("h0", "v", f"h{c0}{AB[x]}", "w"), ("h1", "v", f"h{c1}{AB[x]}", "w")] + e8(x, b, r0), n=3)


def max4(x, b, r0, c0, c1):
    return Op("v_max3_f32 {h1}, {h1}, {e0}, {e1}\n"
              "v_max3_f32 {h0}, {h0}, {e2}, {e3}\n"
              "v_max3_f32 {h1}, {h1}, {e4}, {e5}\n"
              "v_max3_f32 {h0}, {h0}, {e6}, {e7}",
              [("h0", "v", f"h{c0}{AB[x]}", "rw"), ("h1", "v", f"h{c1}{AB[x]}", "rw")] + e8(x, b, r0), n=4)


def max_fin(x):
    # the four chains, then the partner lane (the other 32 keys of the row), then: which lanes' tile max passed the
    # rescale threshold
    return Op("v_max3_f32 {h0}, {h0}, {h1}, {h2}\n"
              "v_max3_f32 {h0}, {h0}, {h3}, {h3}\n"
              "v_mov_b32 {h1}, {h0}\n"
              "s_nop 1\n"
              "v_permlane32_swap_b32 {h0}, {h1}\n"
              "v_max3_f32 {h0}, {h0}, {h1}, {h1}\n"
              "v_cmp_gt_f32 {mk}, {h0}, {thr}",
              [("h0", "v", f"h0{AB[x]}", "rw"), ("h1", "v", f"h1{AB[x]}", "rw"), ("h2", "v", f"h2{AB[x]}", "r"),
               ("h3", "v", f"h3{AB[x]}", "r"), ("mk", "sout", f"mk{AB[x]}", "w"), ("thr", "v", f"thr[{x}]", "r")], n=7)


def vread(i, hi):
    off = (i >> 2) * 4096 + (i & 3) * 256 + (2048 if hi else 0)
    return Op("ds_read_b64_tr_b16 {v}, {va} offset:{off}",
              [("v", "v", f"v{'hi' if hi else 'lo'}[{i}]", "w"), ("va", "v", "va_", "r"),
               ("off", "n", f"VO + {off}", "r")], clobbers=("memory",))


def kread(i):
    return Op(f"ds_read_b128 a[{192 + 4 * i}:{195 + 4 * i}], {{ka}} offset:{{ko}}",
              [("ka", "v", f"W64_KA({i})", "r"), ("ko", "n", f"W64_KOF({i})", "r")], clobbers=("memory",))


def dma_pair(j):
    return Op("s_mov_b32 m0, {lk}\n"
              "s_nop 0\n"
              "global_load_lds_dwordx4 {kofs}, {kb}\n"
              "s_mov_b32 m0, {lv}\n"
              "s_nop 0\n"
              "global_load_lds_dwordx4 {vofs}, {vb}",
              [("lk", "s", f"W64_DMA_LDS_K({j})", "r"), ("kofs", "v", f"W64_DMA_KOFS({j})", "r"),
               ("kb", "s", "run_k", "r"), ("lv", "s", f"W64_DMA_LDS_V({j})", "r"),
               ("vofs", "v", f"W64_DMA_VOFS({j})", "r"), ("vb", "s", "run_v", "r")],
              n=6, clobbers=("memory",))   # M0: reserved, the compiler only ever sets it right before a use


def dma_single(j, which):
    """one K (which = 0) or V (1) piece of the LDS-DMA of tile t+2"""
    kv = "KV"[which]
    return Op("s_mov_b32 m0, {l}\n"
              "s_nop 0\n"
              "global_load_lds_dwordx4 {ofs}, {b}",
              [("l", "s", f"W64_DMA_LDS_{kv}({j})", "r"), ("ofs", "v", f"W64_DMA_{kv}OFS({j})", "r"),
               ("b", "s", "run_v" if which else "run_k", "r")], n=3, clobbers=("memory",))


def dma_m0(j, which):
    return Op("s_mov_b32 m0, {l}", [("l", "s", f"W64_DMA_LDS_{'KV'[which]}({j})", "r")], n=1)


def dma_load(j, which):
    kv = "KV"[which]
    return Op("global_load_lds_dwordx4 {ofs}, {b}",
              [("ofs", "v", f"W64_DMA_{kv}OFS({j})", "r"), ("b", "s", "run_v" if which else "run_k", "r")],
              n=1, clobbers=("memory",))


DMA_LAYOUT = os.environ.get("W64_GEN_DMA", "split")     # pairs: gaps 0..3;  split: K pieces gaps 0..3, V pieces gaps 14..17
LIKELY = os.environ.get("W64_GEN_LIKELY", "0") == "1"   # measured: moving the rescale block out of line made the tile 3.8x slower


def raw(text, clobbers=("memory",)):
    return Op(text, [], n=0, clobbers=clobbers)


# ---- the placement table ------------------------------------------------------------------------------------------
# The compiler's hazard recogniser assumes that an asm statement may have written its outputs with a 16-bit destination
# select, and puts an s_nop in front of the NEXT asm statement if that one touches any of them.  With a running sum, a
# max chain or an exp feeding the next statement that was one wasted issue slot in almost every gap (57 per tile).
# So the rule of the table: a statement never reads or writes a register the statement before it wrote.  (An asm
# statement counts as zero wait states in that look-back, so a register written TWO statements back still draws the pad
# unless a compiler-generated instruction sits in between: 57 -> 31 pads per tile, not 0.)
#   * row max: four chains, two per statement, alternating (r 0..3), combined two gaps after the last update (r 5)
#   * row sums: two accumulators per block, by gap parity
#   * slices: fma / exp / add+cvt+add scheduled per instruction, every consumer at least two gaps behind its producer
#     (an exp may share the gap of its fma: same statement), NT temporaries pairs in flight
# chain timeline r (block A: gap 17 + r, block B: gap 49 + r, wrapping into the next tile):
VLAYOUT = os.environ.get("W64_GEN_VLAYOUT", "burst")   # burst: gaps 32..35 carry five V^T reads each; spread: 1-2 per gap
# slice instructions a gap may take (block A pauses at r 15..18 = gaps 32..35, the V^T read burst; block B runs lighter
# next to the K(t+1) reads of ph4, the LDS-DMA pieces and block A's row max)
if VLAYOUT == "burst":
    CAP_A = {**{r: 5 for r in range(6, 15)}, **{r: 5 for r in range(19, 32)}, **{r: 3 for r in range(32, 42)}}
else:
    CAP_A = {**{r: 4 for r in range(6, 15)}, **{r: 3 for r in range(15, 23)}, **{r: 5 for r in range(23, 32)},
             **{r: 3 for r in range(32, 42)}}
CAP_B = ({**{r: 4 for r in range(6, 15)}, **{r: 3 for r in range(15, 19)}, **{r: 5 for r in range(19, 29)},
          **{r: 3 for r in range(29, 33)}, **{r: 2 for r in range(33, 36)}, 36: 5, **{r: 2 for r in range(37, 42)}}
         if os.environ.get("W64_GEN_BSIZES", "light") == "light" else CAP_A)
FIN_R = 5


def schedule_slices(x, cap, gap0):
    """r -> ops of block x's 16 slices (112 instructions), list-scheduled oldest slice first"""
    if ROWSUM == "dot2":
        deps = {"F0": [], "F1": [], "X0": ["F0"], "X1": ["F1"], "C": ["X0", "X1"], "D": ["C"]}
        rank = {"D": 0, "C": 0, "X0": 1, "X1": 1, "F0": 2, "F1": 2}
        kinds = ("F0", "F1", "X0", "X1", "C", "D")
    else:
        deps = {"F0": [], "F1": [], "X0": ["F0"], "X1": ["F1"], "A0": ["X0"], "C": ["X0", "X1"], "A1": ["X1"]}
        rank = {"A0": 0, "C": 0, "A1": 0, "X0": 1, "X1": 1, "F0": 2, "F1": 2}
        kinds = ("F0", "F1", "X0", "X1", "A0", "C", "A1")
    pending = [(k, o) for k in range(16) for o in kinds]
    place, by_r = {}, {}
    for r in sorted(cap):
        picked = []
        while len(picked) < cap[r]:
            def ready(k, o):
                for d in deps[o]:
                    g = place.get((k, d))
                    if g is None or (g > r - 2 and not (o[0] == "X" and g == r)):
                        return False
                return True
            oldest = min(k for k, _ in pending) if pending else 0
            cands = [(k, o) for (k, o) in pending if ready(k, o) and not (o[0] == "F" and k - oldest >= NT)]
            if not cands:
                break
            k, o = min(cands, key=lambda ko: (rank[ko[1]], ko[0]))
            place[(k, o)] = r
            pending.remove((k, o))
            picked.append((k, o))
        # text order inside the statement: fma, then add / cvt, the exps last (an exp of this gap's fma sits behind it)
        picked.sort(key=lambda ko: {"F": 0, "A": 1, "C": 1, "D": 1, "X": 2}[ko[1][0]])
        parity = (gap0 + r) & 1
        mk = {"F0": lambda k: fma(x, k, 0), "F1": lambda k: fma(x, k, 1), "X0": lambda k: exp(x, k, 0),
              "X1": lambda k: exp(x, k, 1), "A0": lambda k: add(x, k, 0, parity), "A1": lambda k: add(x, k, 1, parity),
              "C": lambda k: cvt(x, k), "D": lambda k: dot2(x, k, parity)}
        by_r[r] = [mk[o](k) for k, o in picked]
    assert not pending, pending
    # the temporaries of slice k are free before slice k + NT starts; P words land two gaps before their MFMA
    for k in range(16 - NT):
        assert max(place[(k, o)] for o in kinds if o[0] in "AC") < place[(k + NT, "F0")], k
    need0 = 48 if x == 0 else 80
    for k in range(16):
        assert gap0 + place[(k, "C")] <= need0 + 4 * (k >> 2) - 2, (x, k)
    return by_r


def chain_ops(x):
    """r -> list of ops for block x"""
    by_r = {0: [max_init(x, 0, 0, 0, 1)], 1: [max_init(x, 0, 8, 2, 3)], 2: [max4(x, 1, 0, 0, 1)], 3: [max4(x, 1, 8, 2, 3)],
            FIN_R: [max_fin(x)]}
    for r, ops in schedule_slices(x, CAP_B if x else CAP_A, 49 if x else 17).items():
        by_r.setdefault(r, []).extend(ops)
    return by_r


DROP = set(filter(None, os.environ.get("W64_GEN_DROP", "").split(",")))   # measurement only: kread vread dma slice max


def build_gaps(variant):
    gaps = build_gaps_full(variant)
    if not DROP:
        return gaps
    def keep(op):
        t = op.text
        if "kread" in DROP and t.startswith("ds_read_b128"): return False
        if "vread" in DROP and t.startswith("ds_read_b64_tr"): return False
        if "dma" in DROP and "global_load_lds" in t: return False
        if "slice" in DROP and t.split()[0] in ("v_fma_f32", "v_exp_f32", "v_add_f32", "@CVT@", "@DOT2@"): return False
        if "max" in DROP and t.startswith("v_max3") : return False
        return True
    return [[op for op in g if keep(op)] for g in gaps]


def build_gaps_full(variant):
    A = chain_ops(0)
    B = chain_ops(1)
    gaps = []
    for g in range(64):
        ops = []
        ph, i = g >> 4, g & 15
        # the MFMA of this gap
        if variant != "DRAIN":
            if ph == 0:
                ops.append(mfma_qk(0, i))
            elif ph == 1:
                if variant == "STEADY":
                    ops.append(mfma_pv(1, i))
            elif ph == 2:
                ops.append(mfma_qk(1, i))
            else:
                ops.append(mfma_pv(0, i))
        elif ph == 1:
            ops.append(mfma_pv(1, i))
        # an LDS-DMA piece of tile t+2: M0 is set right behind the MFMA and the load goes last, so the gap's other
        # instructions are the wait state the M0 write needs (LDS reads do not use M0 on this chip)
        piece = None
        if variant != "DRAIN" and DMA_LAYOUT == "split":
            if g < 4:
                piece = (g, 0)
            if 14 <= g <= 17:
                piece = (g - 14, 1)
        if piece:
            ops.append(dma_m0(*piece))
        nbefore = len(ops)
        # LDS reads first (they land sooner), then the softmax work
        if variant != "DRAIN":
            if VLAYOUT == "burst":
                if 21 <= g <= 31 and g != 22:     # V^T(t) fragments 0..4 (free once ph2 MFMA 16+f has read them); not in block A's FIN gap
                    j = g - 21 if g < 22 else g - 22
                    ops.append(vread(j >> 1, j & 1))
                if 32 <= g <= 35:          # fragments 6..15: five reads per gap (block A's chain idles here)
                    for j in range(5):
                        q = 12 + (g - 32) * 5 + j
                        ops.append(vread(q >> 1, q & 1))
                if g in (36, 37):          # fragment 5
                    ops.append(vread(5, g - 36))
            else:
                # read k (fragment k>>1, half k&1): one per gap from gap 18 (fragment f's registers are free after ph2
                # MFMA 16+f), two per gap in gaps 32..39; the last one goes out in gap 41, six gaps before the wait
                order = list(range(18, 32)) + [g2 for g2 in range(32, 40) for _ in (0, 1)] + [40, 41]
                for k, g2 in enumerate(order):
                    if g2 == g:
                        ops.append(vread(k >> 1, k & 1))
            if ph == 3:                # K(t+1) fragment i (a[192:255] was last read by ph3)
                ops.append(kread(i))
            ops += A.get(g - 17, [])
            if g >= 49:
                ops += B.get(g - 49, [])
        if variant in ("STEADY", "DRAIN") and g + 15 in B and g <= 26:
            ops += B[g + 15]           # block B of the previous tile (r = g + 15)
        if piece:
            if len(ops) == nbefore:
                ops.append(raw("s_nop 0", clobbers=()))
            ops.append(dma_load(*piece))
        if variant != "DRAIN" and DMA_LAYOUT == "pairs" and g < 4:
            ops.append(dma_pair(g))
        gaps.append(ops)
    return gaps


def emit_stmt(ops, out, prefix=None):
    if not ops and not prefix:
        return
    names = {}
    lines = []
    clob = []
    if prefix:
        lines.append(prefix)
        clob.append("memory")
    for op in ops:
        m = {}
        for key, cls, cexpr, acc in op.operands:
            d = names.get(cexpr)
            if d is None:
                d = names[cexpr] = {"name": f"x{len(names)}", "cls": cls, "written": False, "read_first": False}
            if acc in ("r", "rw") and not d["written"]:
                d["read_first"] = True
            if acc in ("w", "rw"):
                d["written"] = True
            m[key] = ("%c[" if cls == "n" else "%[") + d["name"] + "]"
        for ln in op.text.split("\n"):
            lines.append(ln.format(**m))
        for c in op.clobbers:
            if c not in clob:
                clob.append(c)
    outs, ins = [], []
    for cexpr, d in names.items():
        nm = d["name"]
        if d["cls"] == "n":
            ins.append(f'[{nm}] "n"({cexpr})')
        elif d["cls"] == "s":
            ins.append(f'[{nm}] "s"({cexpr})')
        elif d["cls"] == "sout":
            outs.append(f'[{nm}] "=s"({cexpr})')
        elif not d["written"]:
            ins.append(f'[{nm}] "v"({cexpr})')
        elif d["read_first"]:
            outs.append(f'[{nm}] "+v"({cexpr})')
        else:
            outs.append(f'[{nm}] "=&v"({cexpr})')
    assert len(outs) + len(ins) <= 30, (len(outs), len(ins))
    out.append("asm volatile(")
    for k, ln in enumerate(lines):
        ln = ln.replace("@MFMA@", '" W64_MFMA "').replace("@CVT@", '" W64_CVT "').replace("@DOT2@", '" W64_DOT2 "')     # element-type mnemonics: macros of the including kernel
        out.append(f'    "{ln}' + ('\\n\\t"' if k + 1 < len(lines) else '"'))
    out.append("    : " + ", ".join(outs))
    out.append("    : " + ", ".join(ins))
    if clob:
        out.append("    : " + ", ".join(f'"{c}"' for c in clob))
    out[-1] += ");"


def emit_variant(variant, out):
    gaps = build_gaps(variant)
    count = sum(op.n for g in gaps for op in g if not op.text.startswith("@MFMA@"))
    out.append(f"#ifdef W64_GEN_{variant}   /* {count} instructions beside the MFMAs */")
    for g, ops in enumerate(gaps):
        if variant == "DRAIN" and g >= 32:
            break
        prefix = None
        if variant != "DRAIN":
            if g == 0:
                out.append("W64_T(0);")
                prefix = "s_waitcnt lgkmcnt(8)"      # K(t) fragments 0..7 (read in gaps 48..55 of the previous tile)
            if g == 8:
                prefix = "s_waitcnt lgkmcnt(0)"      # fragments 8..15 (no other LDS read since)
            if g == 16:
                out.append("W64_T(1);")
            if g == 32:
                out.append("W64_T(2);")
            if g == 48:
                out.append("W64_T(3);")
                # V^T(t) landed (last read issued in gap 37); tile t+1 landed (the 8 pieces of tile t+2 may fly);
                # the barrier publishes tile t+1 and retires every wave's reads of tile t
                emit_stmt([raw("s_waitcnt lgkmcnt(0)\ns_waitcnt vmcnt(8)\ns_barrier")], out)
                out.append("W64_T(4);")
        nfill = sum(op.n for op in ops if not op.text.startswith("@MFMA@"))
        out.append(f"// gap {g}: {nfill}")
        emit_stmt(ops, out, prefix)
        if variant != "DRAIN":
            if g == 17 + FIN_R:
                out.append(("if (__builtin_expect(mkA != 0, 0))" if LIKELY else "if (mkA != 0)") + " rescale(std::integral_constant<int, 0>{}, h0A);")
                out.append("__builtin_amdgcn_sched_barrier(0);")
            if g == 49 + FIN_R:
                out.append(("if (__builtin_expect(mkB != 0, 0))" if LIKELY else "if (mkB != 0)") + " rescale(std::integral_constant<int, 1>{}, h0B);")
                out.append("__builtin_amdgcn_sched_barrier(0);")
            if g == 63:
                out.append("W64_T(5);")
    out.append(f"#endif  // W64_GEN_{variant}")
    out.append("")


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "duo-attention_amd", "csrc", "duo_prefill_w64_bulk.inc")
    if "-o" in sys.argv:
        dst = sys.argv[sys.argv.index("-o") + 1]
    out = ["// GENERATED by tools/gen_w64_bulk.py — do not edit; the schedule (which instruction rides in which MFMA gap)",
           "// is the table in that script.  Included three times by duo_prefill_w64.h, inside the bulk-tile lambdas.",
           ""]
    for v in ("STEADY", "FIRST", "DRAIN"):
        emit_variant(v, out)
    with open(dst, "w") as f:
        f.write("\n".join(out))
    if "-v" in sys.argv:
        for v in ("STEADY",):
            for g, ops in enumerate(build_gaps(v)):
                print(g, sum(op.n for op in ops if not op.text.startswith("@MFMA@")))


if __name__ == "__main__":
    main()
