"""GPU: the BATCHED entry points of the C ABI (batch row = a grid dimension of the same kernels).

The reference's pools, counters and forward carry a batch dimension (static_kv_cache.py:60-125, llama.py:309-434) and
flash_attn_func batches natively; until round 3 every B > 1 call here was a Python loop of per-row launches.  Checked:
  * the static hot path with B = 2 and 3 (first chunk, later chunks, fused decode steps) — every batch row against the
    oracle run on that row alone, pools bit-equal to running the row alone on the HIP path (data movement and RoPE do not
    depend on the batch), with equal and with per-row different RoPE positions (position_ids[:, 0], llama.py:347-352);
  * op level: batched prefill == the per-row launches bit for bit (same kernel, same tiles), batched decode within the
    attention bar of the oracle; non-contiguous batch strides.
"""
import pytest
import torch

from helpers import ShapeModel, attn_close, heads_from_counts
from oracle.duo_oracle import StaticCacheRef, flash_attn_func_ref, static_forward_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _rand(shape, g):
    return torch.randn(*shape, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("B,starts", [(2, [0, 0]), (3, [0, 0, 0]), (2, [5, 21])])
def test_static_hot_path_batched_rows(B, starts):
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    counts, Hq, Hkv, sink, recent = [1, 3, 0, 4], 16, 4, 16, 48
    heads = heads_from_counts(counts, Hkv)
    steps = (150, 70, 33, 1, 1, 1)
    cap = sum(steps) + 2
    both = DuoAttentionStaticKVCache(ShapeModel(len(counts), Hq, Hkv, D, device=DEV), heads, B, cap, sink, recent)
    solo = [DuoAttentionStaticKVCache(ShapeModel(len(counts), Hq, Hkv, D, device=DEV), heads, 1, cap, sink, recent)
            for _ in range(B)]
    refs = [StaticCacheRef(len(counts), Hkv, D, heads, 1, cap, sink, recent) for _ in range(B)]
    g = torch.Generator().manual_seed(17 + B)
    past = 0
    for S in steps:
        pos = [s + past for s in starts]
        for l in range(len(counts)):
            q, k, v = _rand((B, S, Hq, D), g), _rand((B, S, Hkv, D), g), _rand((B, S, Hkv, D), g)
            out = duo_static_attention_core(q.to(DEV), k.to(DEV), v.to(DEV), both, l, pos if len(set(pos)) > 1 else pos[0],
                                            1.0, 500000.0)
            for b in range(B):
                # (clones: the oracle, like the reference, rotates q and k IN PLACE)
                exp, bud = static_forward_ref(q[b:b + 1].clone(), k[b:b + 1].clone(), v[b:b + 1].clone(), refs[b], l, pos[b],
                                              1.0, 500000.0, round_p=False, out_dtype=torch.float32, return_budget=True)
                attn_close(out[b:b + 1], exp, f"batched B={B} S={S} layer={l} row={b}", bud if S > 1 else None)
                duo_static_attention_core(q[b:b + 1].to(DEV), k[b:b + 1].to(DEV), v[b:b + 1].to(DEV), solo[b], l, pos[b],
                                          1.0, 500000.0)
        past += S
    for l in range(len(counts)):
        n, m = both.kv_seq_len_list[l], both.streaming_kv_seq_len_list[l]
        assert n == past and m == min(past, sink + recent)
        for b in range(B):
            assert (solo[b].kv_seq_len_list[l], solo[b].streaming_kv_seq_len_list[l]) == (n, m)
            for name in ("full_key_states_list", "full_value_states_list"):
                assert torch.equal(getattr(both, name)[l][b, :n], getattr(solo[b], name)[l][0, :n]), (name, l, b)
            for name in ("streaming_key_states_list", "streaming_value_states_list"):
                assert torch.equal(getattr(both, name)[l][b, :m], getattr(solo[b], name)[l][0, :m]), (name, l, b)


def _classes(B, S, group, nf, ns, lenA_f, lenA_s, g, pad=0):
    """random batched inputs; ``pad`` extra rows per batch entry make the batch strides non-contiguous"""
    Hq = (nf + ns) * group
    mk = lambda T, h: _rand((B + pad, h, T, D), g).to(DEV).permute(0, 2, 1, 3)[:B]     # head-major storage, [B, T, h, D] views
    q = _rand((B + pad, S, Hq, D), g).to(DEV)[:B]
    kn, vn = mk(S, nf + ns), mk(S, nf + ns)
    fk, fv = mk(max(lenA_f, 1), max(nf, 1)), mk(max(lenA_f, 1), max(nf, 1))
    sk, sv = mk(max(lenA_s, 1), max(ns, 1)), mk(max(lenA_s, 1), max(ns, 1))
    full = (nf, 0, (fk[:, :lenA_f, :nf], fv[:, :lenA_f, :nf]) if lenA_f else None, (kn[:, :, :nf], vn[:, :, :nf])) if nf else None
    stream = (ns, nf * group, (sk[:, :lenA_s, :ns], sv[:, :lenA_s, :ns]) if lenA_s else None,
              (kn[:, :, nf:], vn[:, :, nf:])) if ns else None
    return q, full, stream


def _row(desc, b):
    if desc is None:
        return None
    n, off, a, bb = desc
    sel = lambda seg: None if seg is None else (seg[0][b], seg[1][b])
    return n, off, sel(a), sel(bb)


@pytest.mark.parametrize("case", [(2, 300, 4, 2, 2, 1000, 64, 0), (3, 257, 4, 1, 3, 70, 64, 1), (2, 64, 2, 3, 0, 513, 0, 2),
                                  (4, 200, 4, 0, 2, 0, 64, 0)])
def test_batched_prefill_equals_per_row_launches(case):
    from duo_attn import _hip
    from duo_attn.backend import HipBackend

    B, S, group, nf, ns, lenA_f, lenA_s, pad = case
    g = torch.Generator().manual_seed(sum(case))
    q, full, stream = _classes(B, S, group, nf, ns, lenA_f, lenA_s, g, pad)
    be = HipBackend()
    scale = D ** -0.5
    out_b = torch.full_like(q, float("nan"))
    out_r = torch.full_like(q, float("nan"))
    _hip.set_debug_flags(256)        # no key-range split: the per-row and the batched launch run the very same tiles
    try:
        be.attention_batched(q, out_b, group, full, stream, scale)
        for b in range(B):
            be.attention(q[b], out_r[b], group, _row(full, b), _row(stream, b), scale)
    finally:
        _hip.set_debug_flags(0)
    assert torch.equal(out_b, out_r)
    # and with the launcher free to split the key range (every batch row has its own partials in the workspace)
    out_s = torch.full_like(q, float("nan"))
    be.attention_batched(q, out_s, group, full, stream, scale)
    for b in range(B):
        for desc in (_row(full, b), _row(stream, b)):
            if desc is None:
                continue
            n, off, a, bb = desc
            kk = torch.cat([t[0].cpu() for t in (a, bb) if t is not None], 0)
            vv = torch.cat([t[1].cpu() for t in (a, bb) if t is not None], 0)
            exact, bud = flash_attn_func_ref(q[b].cpu()[None, :, off:off + n * group], kk[None], vv[None], causal=True,
                                             softmax_scale=scale, round_p=False, out_dtype=torch.float32, return_budget=True)
            attn_close(out_s[b][None, :, off:off + n * group], exact, f"batched prefill {case} row {b}", bud)


@pytest.mark.parametrize("case", [(2, 4, 2, 2, 5000, 64), (3, 4, 1, 3, 40000, 64), (5, 1, 2, 6, 300, 17), (2, 8, 1, 0, 2049, 0)])
def test_batched_decode_rows_match_oracle(case):
    from duo_attn.backend import HipBackend

    B, group, nf, ns, lenA_f, lenA_s = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    q, full, stream = _classes(B, 1, group, nf, ns, lenA_f, lenA_s, g, pad=1)
    be = HipBackend()
    scale = D ** -0.5
    out = torch.full_like(q, float("nan"))
    be.attention_batched(q, out, group, full, stream, scale)
    for b in range(B):
        for desc in (_row(full, b), _row(stream, b)):
            if desc is None:
                continue
            n, off, a, bb = desc
            kk = torch.cat([t[0].cpu() for t in (a, bb) if t is not None], 0)
            vv = torch.cat([t[1].cpu() for t in (a, bb) if t is not None], 0)
            exact = flash_attn_func_ref(q[b].cpu()[None, :, off:off + n * group], kk[None], vv[None], causal=True,
                                        softmax_scale=scale, round_p=False, out_dtype=torch.float32)
            attn_close(out[b][None, :, off:off + n * group], exact, f"batched decode {case} row {b}", None)
