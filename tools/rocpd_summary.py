#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as markdown.

    python tools/rocpd_summary.py gpurun_out/prof/r1_results.db [--pmc] > profiles/<name>.md

Kernel table = the `--stats` view (calls, total, average, min, max per kernel);
with --pmc also the per-kernel mean of every collected counter.
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--top", type=int, default=12)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, avg, mn, mx in rows[: a.top]:
        print(f"| `{short(n)}` | {c} | {s / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / total:.2f} |")
    if a.pmc:
        rows = cur.execute(
            "select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) "
            "from counters_collection group by kernel_name, counter_name order by sum(duration) desc"
        ).fetchall()
        print("\n| kernel | counter | dispatches | mean value | sum | avg dispatch us (profiled) |")
        print("|---|---|---:|---:|---:|---:|")
        for n, cn, c, avg, s, d in rows[: a.top * 4]:
            print(f"| `{short(n)}` | {cn} | {c} | {avg:.4g} | {s:.6g} | {d / 1e3:.2f} |")


if __name__ == "__main__":
    main()
