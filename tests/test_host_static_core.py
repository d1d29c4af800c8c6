"""Host plumbing of the static dual-cache path on CPU, with the oracle plugged in as backend.

Checks the product's control flow (duo_static_attention_core + DuoAttentionStaticKVCache with its
head-major pools, two-segment attention descriptors and in-place streaming update) against the
oracle's procedural restatement of reference llama.py:309-434 — bit-exact, both use the same math.
"""
import pytest
import torch

from helpers import ShapeModel, heads_from_counts
from oracle.duo_oracle import StaticCacheRef, duo_visible_mask, flash_attn_func_ref, static_forward_ref


def _run(counts, Hq, Hkv, chunks, sink, recent, seed=0, decode_steps=3):
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    D = 128
    L = len(counts)
    heads = heads_from_counts(counts, Hkv)
    total = sum(chunks) + decode_steps
    model = ShapeModel(L, Hq, Hkv, D)
    cache = DuoAttentionStaticKVCache(model, heads, 1, total + 5, sink, recent)
    ref = StaticCacheRef(L, Hkv, D, heads, 1, total + 5, sink, recent)
    g = torch.Generator().manual_seed(seed)
    pos = 0
    for S in list(chunks) + [1] * decode_steps:
        for l in range(L):
            q = torch.randn(1, S, Hq, D, generator=g).to(torch.bfloat16)
            k = torch.randn(1, S, Hkv, D, generator=g).to(torch.bfloat16)
            v = torch.randn(1, S, Hkv, D, generator=g).to(torch.bfloat16)
            out = duo_static_attention_core(q.clone(), k.clone(), v.clone(), cache, l, pos, 1.0, 10000.0)
            exp = static_forward_ref(q.clone(), k.clone(), v.clone(), ref, l, pos, 1.0, 10000.0)
            assert torch.equal(out, exp), (S, l)
            n, ns_len = ref.kv_seq_len_list[l], ref.streaming_kv_seq_len_list[l]
            assert cache.kv_seq_len_list[l] == n and cache.streaming_kv_seq_len_list[l] == ns_len
            assert torch.equal(cache.full_key_states_list[l][:, :n], ref.full_key_states_list[l][:, :n])
            assert torch.equal(cache.full_value_states_list[l][:, :n], ref.full_value_states_list[l][:, :n])
            assert torch.equal(cache.streaming_key_states_list[l][:, :ns_len],
                               ref.streaming_key_states_list[l][:, :ns_len])
            assert torch.equal(cache.streaming_value_states_list[l][:, :ns_len],
                               ref.streaming_value_states_list[l][:, :ns_len])
        pos += S
    return cache, ref


@pytest.mark.parametrize(
    "counts,Hq,Hkv,chunks,sink,recent",
    [
        ([1, 2, 0, 4], 8, 4, (7, 9, 20), 4, 8),      # ragged split incl. nf=0 and nf=Hkv
        ([2, 1], 4, 4, (30,), 8, 16),                # MHA, single-shot prefill
        ([1, 3], 8, 4, (5, 5, 5, 5), 2, 3),          # pool saturates mid-way
    ],
)
def test_core_matches_reference_control_flow(oracle_backend, counts, Hq, Hkv, chunks, sink, recent):
    _run(counts, Hq, Hkv, chunks, sink, recent)


def test_evict_last_and_clear(oracle_backend):
    cache, ref = _run([1, 1], 4, 2, (12,), 2, 4, decode_steps=2)
    cache.evict_last(1)
    ref.evict_last(1)
    assert cache.kv_seq_len_list == ref.kv_seq_len_list
    assert cache.streaming_kv_seq_len_list == ref.streaming_kv_seq_len_list
    cache.clear()
    assert cache.kv_seq_len == 0 and cache.streaming_kv_seq_len == 0


def test_overflow_raises(oracle_backend):
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    cache = DuoAttentionStaticKVCache(ShapeModel(1, 2, 2), [[1.0, 0.0]], 1, 4, 1, 2)
    k = torch.zeros(1, 5, 1, 128, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="Trying to put 5 KVs into a cache with max size 4, current size: 0."):
        cache.put_full_kv(0, k, k)


def test_memory_usage_matches_reference_formula(oracle_backend):
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    counts, Hkv, max_size, sink, recent = [1, 3, 0], 4, 50, 4, 8
    cache = DuoAttentionStaticKVCache(ShapeModel(3, 8, Hkv), heads_from_counts(counts, Hkv), 1, max_size,
                                      sink, recent)
    exp = sum(2 * 2 * 128 * (nf * max_size + (Hkv - nf) * (sink + recent)) for nf in counts)
    assert cache.memory_usage == exp
    # reference-visible shapes are token-major
    assert tuple(cache.full_key_states_list[1].shape) == (1, max_size, 3, 128)
    assert tuple(cache.streaming_key_states_list[1].shape) == (1, sink + recent, 1, 128)


@pytest.mark.parametrize("N,S", [(0, 9), (5, 4), (12, 3), (40, 1), (13, 6)])
def test_procedural_semantics_equal_closed_form_mask(N, S):
    """static_forward_ref (procedural: cat, bottom-right causal, compress) == dense attention under the
    closed-form visibility mask of SURVEY §8(a7), for a chunk of S tokens after N cached tokens fed in
    ONE earlier call."""
    D, sink, recent = 128, 4, 8
    g = torch.Generator().manual_seed(1)
    heads = [[1.0, 0.0]]
    ref = StaticCacheRef(1, 2, D, heads, 1, N + S + 1, sink, recent)
    allk, allv = [], []
    outs = None
    pos = 0
    for n in ([N] if N else []) + [S]:
        q = torch.randn(1, n, 4, D, generator=g).to(torch.bfloat16)
        k = torch.randn(1, n, 2, D, generator=g).to(torch.bfloat16)
        v = torch.randn(1, n, 2, D, generator=g).to(torch.bfloat16)
        kk = k.clone()
        outs = static_forward_ref(q, kk, v, ref, 0, pos, 1.0, 10000.0, round_p=False, out_dtype=torch.float32)
        allk.append(kk)   # rotated in place
        allv.append(v)
        pos += n
    K, V = torch.cat(allk, dim=1), torch.cat(allv, dim=1)
    for h, kind in ((0, "full"), (1, "stream")):
        vis = duo_visible_mask(kind, N, S, sink, recent)
        for gq in range(2):
            qh = q[0, :, 2 * h + gq].float()
            s = (qh @ K[0, :, h].float().T) / (D ** 0.5)
            s = s.masked_fill(~vis, float("-inf"))
            exp = torch.softmax(s, -1) @ V[0, :, h].float()
            torch.testing.assert_close(outs[0, :, 2 * h + gq], exp, rtol=1e-5, atol=1e-5)


def test_row_blocks_equal_whole_chunk(oracle_backend):
    """duo_static_attention_row_block: a chunk processed in row blocks (pipeline wavefront) gives the chunk's
    result, not a smaller chunk's — outputs, both pools and all counters — for the first chunk and for later
    chunks, through the streaming window's fill -> slide transition."""
    import torch
    from helpers import ShapeModel, heads_from_counts
    from duo_attn.patch._duo import duo_static_attention_core, duo_static_attention_row_block
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    counts, Hq, Hkv, D, sink, recent = [1, 2, 0, 4], 8, 4, 128, 3, 9
    chunks, blocks = [40, 24, 33], [[16, 16, 8], [8, 16], [33]]
    g = torch.Generator().manual_seed(3)
    mk = lambda S, h: torch.randn(1, S, h, D, generator=g).to(torch.bfloat16)
    data = [[(mk(S, Hq), mk(S, Hkv), mk(S, Hkv)) for _ in counts] for S in chunks]

    def run(by_blocks):
        model = ShapeModel(len(counts), Hq, Hkv, D)
        cache = DuoAttentionStaticKVCache(model, heads_from_counts(counts, Hkv), 1, 128, sink, recent)
        outs = []
        for ci, S in enumerate(chunks):
            pos = cache.kv_seq_len
            for li in range(len(counts)):
                q, k, v = (t.clone() for t in data[ci][li])
                if by_blocks:
                    r0, parts = 0, []
                    for n in blocks[ci]:
                        parts.append(duo_static_attention_row_block(q[:, r0:r0 + n], k[:, r0:r0 + n], v[:, r0:r0 + n],
                                                                    cache, li, r0, S, 1.0, 1e4))
                        r0 += n
                    outs.append(torch.cat(parts, 1))
                else:
                    outs.append(duo_static_attention_core(q, k, v, cache, li, pos, 1.0, 1e4))
        return outs, cache

    whole, c0 = run(False)
    blocked, c1 = run(True)
    for a, b in zip(whole, blocked):
        torch.testing.assert_close(a.float(), b.float(), rtol=0, atol=2 ** -7 * 4)   # bf16 outputs, same math
    assert c0.kv_seq_len_list == c1.kv_seq_len_list and c0.streaming_kv_seq_len_list == c1.streaming_kv_seq_len_list
    for l in range(len(counts)):
        n, m = c0.kv_seq_len_list[l], c0.streaming_kv_seq_len_list[l]
        assert torch.equal(c0.full_key_states_list[l][:, :n], c1.full_key_states_list[l][:, :n])
        assert torch.equal(c0.full_value_states_list[l][:, :n], c1.full_value_states_list[l][:, :n])
        assert torch.equal(c0.streaming_key_states_list[l][:, :m], c1.streaming_key_states_list[l][:, :m])
        assert torch.equal(c0.streaming_value_states_list[l][:, :m], c1.streaming_value_states_list[l][:, :m])


def test_row_blocks_must_come_in_order(oracle_backend):
    import pytest
    import torch
    from helpers import ShapeModel, heads_from_counts
    from duo_attn.patch._duo import duo_static_attention_row_block
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    cache = DuoAttentionStaticKVCache(ShapeModel(1, 4, 2, 128), heads_from_counts([1], 2), 1, 64, 2, 4)
    t = lambda n, h: torch.zeros(1, n, h, 128, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        duo_static_attention_row_block(t(4, 4), t(4, 2), t(4, 2), cache, 0, 4, 16, 1.0, 1e4)   # no block at row 0 yet
    duo_static_attention_row_block(t(4, 4), t(4, 2), t(4, 2), cache, 0, 0, 16, 1.0, 1e4)
    with pytest.raises(ValueError):
        duo_static_attention_row_block(t(4, 4), t(4, 2), t(4, 2), cache, 0, 8, 16, 1.0, 1e4)   # skips rows 4..7


def test_per_row_rope_offsets_in_a_batch(oracle_backend):
    """position_ids rows that start at different positions (a left-padded batch): every batch row is rotated at
    ITS first position, as the reference does by handing position_ids[:, 0] to the RoPE kernel (llama.py:350-352)
    — prefill chunk and fused decode step, equal to running each row alone."""
    from duo_attn.patch._duo import duo_static_attention_core, first_positions
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    D, Hq, Hkv, sink, recent = 128, 4, 2, 2, 4
    heads = heads_from_counts([1], Hkv)
    g = torch.Generator().manual_seed(3)
    mk = lambda B, S, h: torch.randn(B, S, h, D, generator=g).to(torch.bfloat16)
    both = DuoAttentionStaticKVCache(ShapeModel(1, Hq, Hkv, D), heads, 2, 40, sink, recent)
    solo = [DuoAttentionStaticKVCache(ShapeModel(1, Hq, Hkv, D), heads, 1, 40, sink, recent) for _ in range(2)]
    starts = [3, 11]
    for S in (6, 5, 1, 1):
        q, k, v = mk(2, S, Hq), mk(2, S, Hkv), mk(2, S, Hkv)
        past = both.kv_seq_len
        pos = [s + past for s in starts]
        ids = torch.stack([torch.arange(p, p + S) for p in pos])
        assert first_positions(ids) == pos and first_positions(ids[:1].expand(2, S)) == pos[0]
        out = duo_static_attention_core(q.clone(), k.clone(), v.clone(), both, 0, pos, 1.0, 10000.0)
        for b in range(2):
            exp = duo_static_attention_core(q[b:b + 1].clone(), k[b:b + 1].clone(), v[b:b + 1].clone(), solo[b], 0,
                                            pos[b], 1.0, 10000.0)
            assert torch.equal(out[b:b + 1], exp), (S, b)
    for b in range(2):
        n = solo[b].kv_seq_len
        assert torch.equal(both.full_key_states_list[0][b, :n], solo[b].full_key_states_list[0][0, :n])
