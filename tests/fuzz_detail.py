"""Element-level view of the first call of a fuzz case that leaves the bar:  python tests/fuzz_detail.py "<dict>" """
import ast
import sys

import torch

import fuzz_static_path as F
from helpers import ShapeModel, heads_from_counts
from oracle.duo_oracle import StaticCacheRef, static_forward_ref

c = ast.literal_eval(sys.argv[1])
from duo_attn.patch._duo import duo_static_attention_core
from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

Hkv, group, counts, B = c["Hkv"], c["group"], c["counts"], c["B"]
Hq, L, D = Hkv * group, len(counts), 128
heads = heads_from_counts(counts, Hkv)
total = sum(c["chunks"]) + 8
cache = DuoAttentionStaticKVCache(ShapeModel(L, Hq, Hkv, D, device="cuda:0"), heads, B, total, c["sink"], c["recent"])
ref = StaticCacheRef(L, Hkv, D, heads, B, total, c["sink"], c["recent"])
g = torch.Generator().manual_seed(c["seed"])
mk = lambda S, h: (torch.randn((B, S, h, D), generator=g) * c["scale"]).to(torch.bfloat16)
pos = 0
for si, S in enumerate(c["chunks"]):
    for l in range(L):
        q, k, v = mk(S, Hq), mk(S, Hkv), mk(S, Hkv)
        out = duo_static_attention_core(q.cuda(), k.cuda(), v.cuda(), cache, l, pos, c["rope_scale"], c["theta"]).float().cpu()
        exp, bud = static_forward_ref(q, k, v, ref, l, pos, c["rope_scale"], c["theta"], round_p=False, out_dtype=torch.float32,
                                      return_budget=True)
        exp_r, _ = static_forward_ref(q.clone(), k.clone(), v.clone(), StaticCacheRef(L, Hkv, D, heads, B, total, c["sink"], c["recent"]),
                                      l, pos, c["rope_scale"], c["theta"], round_p=True, out_dtype=torch.float32,
                                      return_budget=True) if pos == 0 else (None, None)
        err = (out - exp).abs()
        rms = exp.pow(2).mean().sqrt()
        tol = 1e-3 * exp.abs() + 2.0 ** -8 * exp.abs() + 1e-3 * rms + 2.0 ** -8 * bud
        bad = (err > tol).nonzero()
        print(f"step {si} S={S} layer {l}: {len(bad)} bad of {err.numel()}, rms err/ref {float((out-exp).pow(2).mean().sqrt()/rms):.3e}")
        for idx in bad[:12]:
            i = tuple(idx.tolist())
            extra = f" roundP-oracle {exp_r[i]:+.5f}" if exp_r is not None else ""
            print(f"   {i}: ours {out[i]:+.5f} ref {exp[i]:+.5f} err {err[i]:.5f} tol {tol[i]:.5f} (|ref| part {(1e-3+2**-8)*abs(exp[i]):.5f}, budget {bud[i]:.4f} -> {2**-8*bud[i]:.5f}){extra}")
        if len(bad):
            raise SystemExit(0)
        n, m = ref.kv_seq_len_list[l], ref.streaming_kv_seq_len_list[l]
        ref.full_key_states_list[l][:, :n].copy_(cache.full_key_states_list[l][:, :n].cpu())
        ref.streaming_key_states_list[l][:, :m].copy_(cache.streaming_key_states_list[l][:, :m].cpu())
    pos += S
