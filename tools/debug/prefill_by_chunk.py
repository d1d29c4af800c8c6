"""Per-chunk efficiency of the prefill launches of the bench job (HIP events per launch, summed over the 32 layers)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "duo-attention_amd"))
import torch
import bench
from duo_attn.backend import get_backend

dev = torch.device("cuda", 0)
counts = bench.LLAMA3_8B_FULL_KV_HEADS
ctx, chunk = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
hp = bench.HotPath(counts, (0, 32), ctx, chunk, dev)
be = get_backend()
G, scale = bench.HQ // bench.HKV, bench.D ** -0.5
pf = bench.prefill_flops(counts, ctx, chunk)
cache = hp.cache
for rep in range(2):
    cache.clear()
    rows = []
    for ci, (s, c) in enumerate(hp.chunks):
        evs = []
        for li, nf in enumerate(counts):
            q, k, v = hp.q_c[:, :c], hp.k_c[:, :c], hp.v_c[:, :c]
            fk, fv, sk, sv = cache.split_kv(li, k, v)
            past_l = cache.kv_seq_len_list[li]
            cache.put_full_kv(li, fk, fv)
            out = torch.empty_like(q)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if s == 0:
                e0.record(); be.attention(q[0], out[0], G, (bench.HKV, 0, None, (k[0], v[0])), None, scale); e1.record()
            else:
                ns = bench.HKV - nf
                pk, pv = cache.full_key_states_list[li], cache.full_value_states_list[li]
                ck, cv = cache.get_streaming_kv(li)
                full = (nf, 0, (pk[0, :past_l], pv[0, :past_l]), (pk[0, past_l:past_l + c], pv[0, past_l:past_l + c])) if nf else None
                stream = (ns, nf * G, (ck[0], cv[0]), (sk[0], sv[0])) if ns else None
                e0.record(); be.attention(q[0], out[0], G, full, stream, scale); e1.record()
            cache.update_streaming_kv(li, sk, sv)
            evs.append((e0, e1, nf))
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b, _ in evs) * 1e-3
        rows.append((s, sum(pf[ci]), t))
    if rep == 1:
        for s, f, t in rows:
            print(f"chunk at {s:7d}: {f / 1e12:8.1f} TFLOP in {t * 1e3:8.1f} ms = {f / t / 1e12:7.1f} TFLOP/s ({f / t / 2.5e15:.3f})")
        F, T = sum(r[1] for r in rows), sum(r[2] for r in rows)
        print(f"total: {F / T / 1e12:.1f} TFLOP/s ({F / T / 2.5e15:.3f})")
