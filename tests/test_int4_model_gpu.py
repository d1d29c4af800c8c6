"""GPU: a HuggingFace Llama driven through the INT4 dual KV cache (duo_attn/patch/int4.py — the attention flow of
reference demo/w8a8kv4_llama.py:174-287 with fp16 linears): chunked prefill + decode with a streaming window that
evicts.  Every cache operation the model performs is recorded and replayed against the INT4 oracle (pinned to the
reference's own kernel, tests/test_int4_golden.py): the pools hold exactly quantize_int4_ref(rotated K / V rows), and
every attention output equals exact attention over the oracle-dequantised pools.  End to end, the logits stay within the
quantisation error of the un-quantised fp16 model when nothing is evicted."""
import copy

import numpy as np
import pytest
import torch

from oracle.duo_oracle import flash_attn_func_ref
from oracle.int4_oracle import dequantize_int4_ref, quantize_int4_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HEADS = np.array([[1.0, 0.0], [0.0, 0.0], [1.0, 1.0]])


def tiny16(seed=0):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128, vocab_size=211, max_position_embeddings=8192,
                      rope_theta=500000.0, attn_implementation="eager", tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).to(torch.float16).eval().to(DEV)


def _close16(ours, ref, bud, what):
    o, r = ours.float().cpu(), ref.float()
    err = (o - r).abs()
    tol = 1e-3 * r.abs() + 2.0 ** -10 * r.abs() + 2.0 ** -10 * bud + 1e-3 * r.pow(2).mean().sqrt()
    assert torch.isfinite(o).all() and (err <= tol).all(), f"{what}: max err {err.max():.3e}"


def test_int4_model_matches_oracle_and_fp16_model():
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache
    from duo_attn.patch.int4 import enable_llama_duo_attention_int4_kv_eval

    sink, recent, chunk = 8, 24, 64
    log = []

    class Rec(DuoAttentionStaticINT4KVCache):
        def put(self, l, k, v, dequantize=True):
            log.append(("put", l, k.detach().cpu().clone(), v.detach().cpu().clone()))
            return super().put(l, k, v, dequantize)

        def prefill_attention(self, l, q, k, v, scale=None):
            out = super().prefill_attention(l, q, k, v, scale)
            log.append(("attn", l, q.detach().cpu().clone(), out.detach().cpu().clone()))
            return out

        def decode_attention(self, l, q, scale=None):
            out = super().decode_attention(l, q, scale)
            log.append(("attn", l, q.detach().cpu().clone(), out.detach().cpu().clone()))
            return out

        def compress(self, l):
            super().compress(l)
            log.append(("compress", l))

    base = tiny16(seed=21)
    model = copy.deepcopy(base)
    enable_llama_duo_attention_int4_kv_eval(model, HEADS.copy())
    N = 153
    kv = Rec(model, HEADS, 1, N + 8, sink, recent, chunk)
    ids = torch.randint(0, 211, (1, N), generator=torch.Generator().manual_seed(22)).to(DEV)
    steps = [64, 64, 22, 1, 1, 1]
    pos = 0
    with torch.no_grad():
        for c in steps:
            out = model(input_ids=ids[:, pos:pos + c], past_key_values=kv, use_cache=True)
            pos += c
    assert kv.kv_seq_len == sum(steps)
    assert out.logits.shape == (1, 1, 211) and torch.isfinite(out.logits).all()
    # the streaming pools were compacted to sink + recent rows (layers that have streaming heads)
    assert kv.streaming_kv_seq_len_list[0] == sink + recent and kv.streaming_kv_seq_len_list[1] == sink + recent

    # ---- replay the recorded cache traffic on the oracle ------------------------------------------------------
    L, Hkv, G = 3, 2, 2
    nf = [int(h.sum()) for h in HEADS]
    pools = [{"fk": [], "fv": [], "sk": [], "sv": []} for _ in range(L)]      # lists of (packed, s, z) row blocks
    last_put = {}

    def deq(blocks):     # -> [T, h, 128] fp32
        if not blocks:
            return None
        p = np.concatenate([b[0] for b in blocks]); s = np.concatenate([b[1] for b in blocks]); z = np.concatenate([b[2] for b in blocks])
        return torch.from_numpy(dequantize_int4_ref(p, s, z).astype(np.float32))

    def rows(blocks):
        return sum(b[0].shape[0] for b in blocks)

    n_checked = 0
    for ev in log:
        if ev[0] == "put":
            _, l, k, v = ev
            first = rows(pools[l]["fk"]) + rows(pools[l]["sk"]) == 0 and not last_put.get(("seen", l))
            last_put[l] = (k, v, first)
            last_put[("seen", l)] = True
            for name, src, lo, hi in (("fk", k, 0, nf[l]), ("fv", v, 0, nf[l]), ("sk", k, nf[l], Hkv), ("sv", v, nf[l], Hkv)):
                if hi > lo:
                    pools[l][name].append(quantize_int4_ref(src[0, :, lo:hi].numpy()))
        elif ev[0] == "attn":
            _, l, q, out = ev
            k, v, first = last_put[l]
            S = q.shape[1]
            if first:       # first chunk: every head causal over the chunk's own un-quantised K/V
                ref, bud = flash_attn_func_ref(q, k, v, round_p=False, out_dtype=torch.float32, return_budget=True)
            else:
                refs, buds = [], []
                for name_k, name_v, lo, hi in (("fk", "fv", 0, nf[l] * G), ("sk", "sv", nf[l] * G, Hkv * G)):
                    if hi > lo:
                        r, b = flash_attn_func_ref(q[:, :, lo:hi], deq(pools[l][name_k])[None], deq(pools[l][name_v])[None],
                                                   round_p=False, out_dtype=torch.float32, return_budget=True)
                        refs.append(r); buds.append(b)
                ref, bud = torch.cat(refs, 2), torch.cat(buds, 2)
            _close16(out, ref, bud if S > 1 else 0 * bud, f"layer {l} S={S}")
            n_checked += 1
        else:
            l = ev[1]
            if nf[l] < Hkv:
                for name in ("sk", "sv"):
                    p = np.concatenate([b[0] for b in pools[l][name]]); s = np.concatenate([b[1] for b in pools[l][name]])
                    z = np.concatenate([b[2] for b in pools[l][name]])
                    if p.shape[0] > sink + recent:
                        keep = np.r_[0:sink, p.shape[0] - recent:p.shape[0]]
                        p, s, z = p[keep], s[keep], z[keep]
                    pools[l][name] = [(p, s, z)]
    assert n_checked == L * len(steps)
    # the device pools hold exactly the oracle's codes / scales / zero points
    for l in range(L):
        for name, cache in (("fk", kv.full_key_caches[l]), ("fv", kv.full_value_caches[l]),
                            ("sk", kv.streaming_key_caches[l]), ("sv", kv.streaming_value_caches[l])):
            if not pools[l][name]:
                continue
            p = np.concatenate([b[0] for b in pools[l][name]]); s = np.concatenate([b[1] for b in pools[l][name]])
            z = np.concatenate([b[2] for b in pools[l][name]])
            T = p.shape[0]
            assert np.array_equal(cache.quantized_data[0, :T].cpu().numpy(), p), (l, name)
            sz = cache.scale_zero[0, :T].cpu().numpy().view(np.uint16)
            assert np.array_equal(sz[..., 0], s.view(np.uint16)) and np.array_equal(sz[..., 1], z.view(np.uint16)), (l, name)

    # ---- end to end: a window that covers the context -> only the 4-bit quantisation separates it from fp16 ----
    model2 = copy.deepcopy(base)
    enable_llama_duo_attention_int4_kv_eval(model2, HEADS.copy())
    kv2 = DuoAttentionStaticINT4KVCache(model2, HEADS, 1, N + 8, 128, 128, chunk)
    pos = 0
    with torch.no_grad():
        for c in steps:
            o2 = model2(input_ids=ids[:, pos:pos + c], past_key_values=kv2, use_cache=True)
            pos += c
            want = base(input_ids=ids[:, :pos]).logits[:, -1:, :]
            rel = ((o2.logits.float() - want.float()).norm() / want.float().norm()).item()
            assert rel < 0.12, (c, rel)      # int4 K/V (step = range / 15 per row) through 3 layers


def test_int4_enabler_refuses_bf16():
    from transformers import LlamaConfig, LlamaForCausalLM

    from duo_attn.patch.int4 import enable_llama_duo_attention_int4_kv_eval

    cfg = LlamaConfig(hidden_size=256, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                      num_key_value_heads=2, head_dim=128, vocab_size=50)
    m = LlamaForCausalLM(cfg).to(torch.bfloat16).to(DEV)
    with pytest.raises(ValueError, match="fp16"):
        enable_llama_duo_attention_int4_kv_eval(m, np.ones((1, 2)))
