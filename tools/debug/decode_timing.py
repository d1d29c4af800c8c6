"""Per-workgroup s_memtime timeline of ONE decode scan launch (needs a -DDUO_DECODE_TIMING build:
SRC=duo_decode tools/debug/build_variant.sh dtiming -DDUO_DECODE_TIMING;
DUO_ATTN_HIP_LIB=.../lib_dtiming.so python tools/debug/decode_timing.py [nf])"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "duo-attention_amd"))
import numpy as np
import torch
from duo_attn import _hip

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 512: the long-prologue scan kernel
HQ, HKV, D, N, W = 32, 8, 128, 131072, 384
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
ns = HKV - nf
q = torch.randn(HQ, D, generator=g, device=dev).to(torch.bfloat16)
k = torch.randn(HKV, D, generator=g, device=dev).to(torch.bfloat16)
v = torch.randn(HKV, D, generator=g, device=dev).to(torch.bfloat16)
out = torch.empty_like(q)
mk = lambda h, T: torch.randn(h, T, D, generator=g, device=dev).to(torch.bfloat16).permute(1, 0, 2)
fk, fv = mk(max(nf, 1), N + 8), mk(max(nf, 1), N + 8)
sk, sv = mk(max(ns, 1), W + 1), mk(max(ns, 1), W + 1)
lib = _hip.load_library()
_hip.set_debug_flags(flags)
for it in range(5):
    _hip.decode_layer(q, k, v, out, nf, fk[:, :nf], fv[:, :nf], N, sk[:, :ns], sv[:, :ns], W, 128, 256, N, 1.0, 5e5, D ** -0.5)
torch.cuda.synchronize()
buf = np.zeros((2048, 12), dtype=np.uint64)
rc = lib.duo_debug_decode_timing(buf.ctypes.data_as(ctypes.c_void_p))
idx = np.nonzero(buf[:, 0] > 0)[0]
t = buf[idx].astype(np.int64)
loop = t[:, 2] - t[:, 1]
heavy = loop > 0.5 * np.median(loop)
per_xcd = {int(x): [int(np.median(loop[heavy & (idx % 8 == x)])), int(loop[heavy & (idx % 8 == x)].min()), int(loop[heavy & (idx % 8 == x)].max())] for x in range(8)}
order = np.argsort(-loop)
print("per XCD (block id % 8) median/min/max loop cycles of the long workgroups:", per_xcd)
print("slowest 12 workgroups (block id, loop cycles):", [(int(idx[i]), int(loop[i])) for i in order[:12]])
print("fastest 6 long workgroups:", [(int(idx[i]), int(loop[i])) for i in order[::-1] if heavy[i]][:6])
t0 = t[:, 0].min()
t = t - t0
MHZ = 100.0   # s_memtime ticks at the constant 100 MHz reference on this chip? reported raw; see the spread
print("flags", flags, "prologue split (median cycles): entry->addresses formed / first loads issued", int(np.median(t[:, 4] - t[:, 0])), " ->RoPE factors", int(np.median(t[:, 5] - t[:, 4])), " ->q rotated", int(np.median(t[:, 1] - t[:, 5])))
print(json.dumps({"rc": rc, "nf": nf, "workgroups": int(len(t)),
                  "start_first_last": [int(t[:, 0].min()), int(t[:, 0].max())],
                  "prologue_end_median": int(np.median(t[:, 1])), "prologue_median": int(np.median(t[:, 1] - t[:, 0])),
                  "loop_median": int(np.median(t[:, 2] - t[:, 1])), "loop_min_max": [int((t[:, 2] - t[:, 1]).min()), int((t[:, 2] - t[:, 1]).max())],
                  "loop_end_first_median_last": [int(t[:, 2].min()), int(np.median(t[:, 2])), int(t[:, 2].max())],
                  "end_last": int(t[:, 3].max()), "epilogue_median": int(np.median(t[:, 3] - t[:, 2]))}))

# ---- constant-rate clock (s_memrealtime, 100 MHz = 10 ns ticks): the only times that compare across XCDs ----
rt0, rt1 = buf[idx, 8].astype(np.int64), buf[idx, 9].astype(np.int64)
base = rt0.min()
xcd = idx % 8
print("realtime (10 ns ticks from the first workgroup's entry): launch span", int(rt1.max() - base),
      " entry spread", int(rt0.max() - base))
print("per XCD: last workgroup end / median end of the long workgroups / median duration:",
      {int(x): [int((rt1[xcd == x]).max() - base), int(np.median(rt1[(xcd == x) & heavy]) - base),
                int(np.median((rt1 - rt0)[(xcd == x) & heavy]))] for x in range(8)})
t = buf[idx].astype(np.int64)
print("epilogue split (median cycles): loop end -> wave groups combined", int(np.median(t[:, 6] - t[:, 2])),
      " -> LDS written + barrier", int(np.median(t[:, 7] - t[:, 6])), " -> partial stored", int(np.median(t[:, 3] - t[:, 7])))
