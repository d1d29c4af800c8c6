"""s_memtime anatomy of the w64 prefill kernel's bulk tile loop (needs a -DW64_TIMING build, tools/debug/build_variant.sh):
DUO_ATTN_HIP_LIB=.../lib_timing.so python tools/debug/w64_timing.py"""
import ctypes, json, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "duo-attention_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench_kernels

sys.argv = ["bench_kernels.py", "prefill", "--nf", "4", "--past", "65536", "--chunk", "16384", "--reps", "2"]
bench_kernels.main()
torch.cuda.synchronize()
from duo_attn import _hip
lib = _hip.load_library()
buf = (ctypes.c_uint32 * 8)()
rc = lib.duo_debug_w64_timing(buf)
v = list(buf)
n = max(v[6], 1)
names = ["P4 end -> P1 start (K wait, loop)", "P1 (QK_A + V reads)", "P2 (QK_B + softmax A)", "P3 (PV_A + softmax A/B)",
         "vmcnt(0) + barrier", "P4 (PV_B + K reads + softmax B + DMA)"]
print(json.dumps({"rc": rc, "tiles": v[6], "cycles_per_tile": {names[k]: round(v[k] / n, 1) for k in range(6)},
                  "sum": round(sum(v[:6]) / n, 1)}, indent=1))
