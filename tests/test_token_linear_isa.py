"""CPU: audit of the built code of csrc/duo_linear.hip.

The kernel's stream loads (weight groups, the prologue's token-row chunks) are inline asm, outside the compiler's wait
bookkeeping, with explicit counted ``s_waitcnt vmcnt(N)``.  That is only correct if NOTHING reads or writes a load's
destination registers between its issue and the wait that covers it — the compiler knows of no hazard there, so a
register copy at a control-flow merge or a live-range split would silently move garbage.  This test compiles the file
for gfx950 and simulates the in-order vector-memory queue over every instantiation's instruction stream (loop bodies
twice, so loop-carried state is covered): every instruction that touches a register with an outstanding asm load is a
failure.  (The GPU parity tests check results; this checks the property the results depend on, for every
instantiation, including the ones no parity case happens to exercise.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "duo-attention_amd", "csrc", "duo_linear.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(tok):
    """VGPR numbers named by one operand token: v12, v[4:7]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _operand_regs(ins):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", ins):
        out |= _regs(tok)
    return out


def _kernels(asm_text):
    for m in re.finditer(r"^(_ZN\S*duo_token_linear_kernel[^\s:]*):[^\n]*\n(.*?)\n\.Lfunc_end", asm_text, flags=re.S | re.M):
        yield m.group(1), m.group(2)


def _audit(body, has_rows=None):
    """returns (violations, number of asm loads seen).  ``has_rows``: the kernel's asm statements branch on ONE wave-uniform
    flag (`s_cmp_eq_u32 flag, 0` + `s_cbranch_scc1 Nf` = taken when the wave has no rows, `s_cbranch_scc0 Nf` = taken when it
    has); True / False fixes those branches for the whole walk (the two kinds of wave), None explores both ways at each."""
    prog = []          # (kind, text, in_asm)   kind: 'label' | 'ins'
    in_asm = False
    for raw in body.splitlines():
        if "ASMSTART" in raw:
            in_asm = True
            continue
        if "ASMEND" in raw:
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            if re.fullmatch(r"\.LBB\d+_\d+:", line or ""):
                prog.append(("label", line[:-1], False))
            continue
        if re.fullmatch(r"\d+:", line):
            prog.append(("label", f"{line[:-1]}@{len(prog)}", False))      # numeric local label of an asm statement
            continue
        if line.endswith(":"):
            prog.append(("label", line[:-1], False))
            continue
        prog.append(("ins", line, in_asm))
    labels = {t: i for i, (k, t, _) in enumerate(prog) if k == "label"}
    violations, asm_loads = [], set()

    def step(i, text, is_asm, queue):
        """one instruction on one path; queue = the in-order vector-memory queue, tuple of (is_asm_load, dst regs)"""
        op = text.split()[0]
        if op == "s_waitcnt":
            vm = re.search(r"vmcnt\((\d+)\)", text)
            if vm:
                keep = int(vm.group(1))
                queue = queue[max(0, len(queue) - keep):] if keep else ()
            return queue
        touched = _operand_regs(text)
        busy = set().union(*[d for a, d in queue if a]) if queue else set()
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            ops = text[len(op):].split(",")
            dst = frozenset(_regs(ops[0].strip()))
            src = _operand_regs(",".join(ops[1:]))
            if src & busy:
                violations.append(f"address of `{text}` reads {sorted(src & busy)} with a load outstanding")
            if is_asm:
                asm_loads.add(i)
            return queue + ((is_asm, dst),)
        if touched & busy:
            violations.append(f"`{text}` touches {sorted(touched & busy)} with a load outstanding")
        if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
            return queue + ((False, frozenset()),)
        return queue

    # every path through the control-flow graph (conditional branches fork; a (position, queue) pair is visited once)
    seen, work = set(), [(0, ())]
    while work:
        i, queue = work.pop()
        while i < len(prog):
            if (i, queue) in seen:
                break
            seen.add((i, queue))
            kind, text, is_asm = prog[i]
            if kind == "label":
                i += 1
                continue
            if text.startswith("s_endpgm"):
                break
            m = re.match(r"(s_branch|s_cbranch\S*)\s+(\.LBB\d+_\d+|\d+[fb])", text)
            if m:
                name = m.group(2)
                if name[0] == ".":
                    tgt = labels[name]
                else:       # numeric local label: the next (f) / previous (b) definition of that number
                    hits = [j for j, (k2, t2, _) in enumerate(prog) if k2 == "label" and t2.split("@")[0] == name[:-1]]
                    tgt = min(j for j in hits if j > i) if name[-1] == "f" else max(j for j in hits if j < i)
                if m.group(1) == "s_branch":
                    i = tgt
                    continue
                if name[0] != "." and has_rows is not None and m.group(1) in ("s_cbranch_scc1", "s_cbranch_scc0"):
                    taken = (not has_rows) if m.group(1) == "s_cbranch_scc1" else has_rows
                    i = tgt if taken else i + 1
                    continue
                work.append((tgt, queue))
                i += 1
                continue
            queue = step(i, text, is_asm, queue)
            i += 1
        if len(seen) > 2_000_000:
            raise RuntimeError("audit: state space exploded")
    return sorted(set(violations)), len(asm_loads)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_no_instruction_touches_a_register_with_an_asm_load_outstanding(tmp_path):
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps", "-c", SRC, "-o", "l.o"],
                   check=True, cwd=tmp_path, capture_output=True)
    text = open(tmp_path / "duo_linear-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    names = []
    for name, body in _kernels(text):
        for has_rows in (True, False):          # a wave that streams rows / a wave that only stages the token rows
            v, n_loads = _audit(body, has_rows)
            assert n_loads >= (10 if has_rows else 2), f"{name}: the audit did not see the asm loads ({n_loads})"
            assert not v, f"{name} (has_rows={has_rows}):\n  " + "\n  ".join(v[:12])
        names.append(name)
    assert len(names) == 12, names          # 4 row counts x 3 prologues


def test_audit_catches_a_copy_before_the_wait():
    """the checker itself: a v_mov of a destination register before the covering wait is reported, after it is not"""
    bad = """
    ;ASMSTART
    global_load_dwordx4 v[10:13], v1, s[2:3]
    ;ASMEND
    v_mov_b32 v20, v11
    s_waitcnt vmcnt(0)
    """
    good = bad.replace("v_mov_b32 v20, v11\n    s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(0)\n    v_mov_b32 v20, v11")
    assert _audit(bad)[0] and not _audit(good)[0]
    counted = """
    ;ASMSTART
    global_load_dwordx4 v[10:13], v1, s[2:3]
    ;ASMEND
    ;ASMSTART
    global_load_dwordx4 v[14:17], v1, s[2:3]
    ;ASMEND
    s_waitcnt vmcnt(1)
    v_add_f32 v0, v10, v11
    v_add_f32 v0, v14, v0
    """
    v, _ = _audit(counted)
    assert len(v) == 1 and "v14" in v[0]
    # a branch over the loads inside an asm statement is a path of its own: the wait that is enough behind the loads is not
    # enough for an OLDER load on the path that skipped them
    skipping = """
    ;ASMSTART
    global_load_dwordx4 v[20:23], v1, s[2:3]
    ;ASMEND
    ;ASMSTART
    s_cmp_eq_u32 s9, 0
    s_cbranch_scc1 1f
    global_load_dwordx4 v[10:13], v1, s[2:3]
    1:
    ;ASMEND
    ;ASMSTART
    s_waitcnt vmcnt(1)
    ;ASMEND
    v_add_f32 v0, v20, v21
    """
    assert _audit(skipping, has_rows=False)[0] and not _audit(skipping, has_rows=True)[0]
    fixed = skipping.replace("s_waitcnt vmcnt(1)\n", "s_waitcnt vmcnt(1)\n    s_cmp_eq_u32 s9, 0\n    s_cbranch_scc0 2f\n    s_waitcnt vmcnt(0)\n    2:\n")
    assert not _audit(fixed, has_rows=False)[0] and not _audit(fixed, has_rows=True)[0]
