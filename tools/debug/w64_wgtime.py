"""Life of every workgroup of ONE prefill launch (needs the -DW64_WGTIME build: tools/debug/build_variant.sh wgtime -DW64_WGTIME;
DUO_ATTN_HIP_LIB=duo-attention_amd/lib/ab/lib_wgtime.so python tools/debug/w64_wgtime.py --rows 2048 --nf 4 --past 114688).

Per workgroup, on the constant 100 MHz clock: entry -> Q fragments ready -> prologue tiles landed -> tile loops done -> stores
drained, the CU it ran on (XCC, SE, CU from HW_ID / XCC_ID) and the tiles it walked.  Printed: the fixed cost of a workgroup by
phase, the per-tile time from a regression over the workgroups, and the idle gaps between consecutive workgroups of a CU."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "debug"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--nf", type=int, default=4)
    ap.add_argument("--past", type=int, default=114688)
    ap.add_argument("--r1", type=int, default=16384)
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    import prefill_launch_map as plm
    from duo_attn import _hip

    b = plm.Bench(a.past, max(a.r1, a.rows))
    _hip.set_debug_flags(a.flags)
    for _ in range(2):
        b.launch(a.rows, a.nf, a.past, a.r1)
    torch.cuda.synchronize()
    plan = _hip.last_prefill_plan()
    info = _hip.prefill_plan(a.nf, 8 - a.nf, 4, a.rows, a.past, a.r1, min(a.past, 384), a.r1, force=plan[0] | (plan[1] << 8))
    n = min(info["blocks"], 8192)
    lib = _hip.load_library()
    buf = (ctypes.c_ulonglong * (8 * n))()
    lib.duo_debug_w64_wgtime.restype = ctypes.c_int
    rc = lib.duo_debug_w64_wgtime(buf, n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    live = t[:, 4] > 0
    t = t[live]
    us = lambda x: x / 100.0
    t0 = t[:, 0].min()
    tiles = (t[:, 7] >> 32).astype(np.int64)
    bulk = (t[:, 7] & 0xffff).astype(np.int64) + ((t[:, 7] >> 16) & 0xffff).astype(np.int64)
    cu = ((t[:, 6] & 0xf) << 8) | (((t[:, 5] >> 13) & 0x7) << 5) | (((t[:, 5] >> 12) & 1) << 4) | ((t[:, 5] >> 8) & 0xf)
    ph = {"entry -> Q ready": us(t[:, 1] - t[:, 0]), "Q ready -> prologue tiles landed": us(t[:, 2] - t[:, 1]),
          "tile loops": us(t[:, 3] - t[:, 2]), "epilogue (stores drained)": us(t[:, 4] - t[:, 3])}
    print(json.dumps({"rc": rc, "plan": plan[:2], "workgroups": int(live.sum()), "distinct CUs": int(len(set(cu.tolist()))),
                      "launch span us": float(us(t[:, 4].max() - t0))}))
    for k, v in ph.items():
        print(f"  {k:36s} mean {v.mean():8.2f}  p10 {np.percentile(v, 10):8.2f}  p90 {np.percentile(v, 90):8.2f} us")
    loops = ph["tile loops"]
    A = np.stack([tiles - bulk, bulk, np.ones_like(tiles)], 1).astype(float)
    coef, *_ = np.linalg.lstsq(A, loops, rcond=None)
    print(f"  tile loops ~ {coef[0]:.3f} us x general tiles + {coef[1]:.3f} us x bulk tiles + {coef[2]:.2f} us   "
          f"(general tiles per workgroup: mean {np.mean(tiles - bulk):.1f})")
    # gaps between consecutive workgroups of one CU
    gaps = []
    for c in set(cu.tolist()):
        m = np.where(cu == c)[0]
        order = m[np.argsort(t[m, 0])]
        for i, j in zip(order[:-1], order[1:]):
            gaps.append(us(t[j, 0] - t[i, 4]))
    if gaps:
        g = np.array(gaps)
        print(f"  idle between consecutive workgroups of a CU: mean {g.mean():.2f}  p10 {np.percentile(g, 10):.2f}  p90 {np.percentile(g, 90):.2f} us  ({len(g)} gaps)")
    first = us(t[:, 0] - t0)
    print(f"  first-round entry spread: p50 {np.percentile(first[first < 50], 50):.2f}  max {first[first < 50].max():.2f} us")


if __name__ == "__main__":
    main()
