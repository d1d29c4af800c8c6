"""INT4 KV pools: oracle self-checks on CPU; on the GPU the HIP kernels are bit-exact against the
oracle (quantise / dequantise / compaction) and the fused decode attention matches attention over the
oracle's dequantised pools."""
import numpy as np
import pytest
import torch

from helpers import ShapeModel, heads_from_counts
from oracle.int4_oracle import dequantize_int4_ref, quantize_int4_ref, roundf_ref

DEV = "cuda:0"


# ----------------------------------------------------------------------------- CPU: the oracle itself
def test_roundf_is_half_away_from_zero():
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.49999997, 14.5, 15.4999], dtype=np.float32)
    assert roundf_ref(x).tolist() == [1.0, 2.0, 3.0, -1.0, -2.0, 0.0, 15.0, 15.0]


def test_oracle_layout_and_error_bound():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((7, 3, 128)).astype(np.float16)
    p, s, z = quantize_int4_ref(x)
    assert p.shape == (7, 3, 64) and p.dtype == np.uint8 and s.dtype == np.float16
    # zero point = row min, codes span 0..15, even element in the high nibble
    assert np.array_equal(z, x.min(-1))
    i = x[0, 0].astype(np.float32).argmax()
    code = (p[0, 0, i // 2] >> 4) if i % 2 == 0 else (p[0, 0, i // 2] & 15)
    assert code == 15
    d = dequantize_int4_ref(p, s, z).astype(np.float32)
    step = (x.astype(np.float32).max(-1) - x.astype(np.float32).min(-1)) / 15
    assert (np.abs(d - x.astype(np.float32)) <= step[..., None] * 0.5 + 4e-3).all()
    # constant row: scale = 1e-8 -> fp16 0, every code 0, dequantises to the constant
    c = np.full((1, 1, 128), 1.5, dtype=np.float16)
    p, s, z = quantize_int4_ref(c)
    assert (p == 0).all() and s[0, 0] == 0 and (dequantize_int4_ref(p, s, z) == 1.5).all()


# ----------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def _pools(T, h, head_major=True):
    if head_major:
        q = torch.zeros(h, T, 64, dtype=torch.uint8, device=DEV).permute(1, 0, 2)
        sz = torch.zeros(h, T, 2, dtype=torch.float16, device=DEV).permute(1, 0, 2)
    else:
        q = torch.zeros(T, h, 64, dtype=torch.uint8, device=DEV)
        sz = torch.zeros(T, h, 2, dtype=torch.float16, device=DEV)
    return q, sz


@gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("head_major", [True, False])
@pytest.mark.parametrize("S,h,row0", [(1, 1, 0), (37, 3, 5), (300, 8, 17)])
def test_quantize_bit_exact(dtype, head_major, S, h, row0):
    from duo_attn import _hip

    g = torch.Generator().manual_seed(S)
    x = (torch.randn(S, h + 2, 128, generator=g) * 2).to(dtype)
    x[0, 1] = 0.75                                   # constant row
    if S > 3:
        x[3, 1, :64] = 10.0                          # two-level row: every code is 0 or 15
        x[3, 1, 64:] = -3.0
    q, sz = _pools(row0 + S + 2, h, head_major)
    _hip.int4_quantize(x.to(DEV)[:, 1:1 + h], q, sz, row0)
    xf = x[:, 1:1 + h].float().numpy()
    p, s, z = quantize_int4_ref(xf)
    assert np.array_equal(q.cpu().numpy()[row0:row0 + S], p)
    assert np.array_equal(sz.cpu()[row0:row0 + S, :, 0].numpy(), s)
    assert np.array_equal(sz.cpu()[row0:row0 + S, :, 1].numpy(), z)
    assert (q.cpu().numpy()[:row0] == 0).all() and (q.cpu().numpy()[row0 + S:] == 0).all()


@gpu
@pytest.mark.parametrize("T,h", [(1, 1), (50, 3), (1000, 4)])
def test_dequantize_bit_exact(T, h):
    from duo_attn import _hip

    rng = np.random.default_rng(T)
    p = rng.integers(0, 256, (T, h, 64), dtype=np.uint8)
    s = (rng.random((T, h)) * 0.5).astype(np.float16)
    z = rng.standard_normal((T, h)).astype(np.float16)
    q, sz = _pools(T, h)
    q.copy_(torch.from_numpy(p))
    sz[..., 0].copy_(torch.from_numpy(s))
    sz[..., 1].copy_(torch.from_numpy(z))
    out = torch.empty(T * h * 128, dtype=torch.float16, device=DEV)
    got = _hip.int4_dequantize(q, sz, T, out).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), dequantize_int4_ref(p, s, z).view(np.uint16))


@gpu
@pytest.mark.parametrize("sink,recent,length", [(128, 256, 385), (128, 256, 384 + 4096), (4, 8, 13), (4, 8, 12), (16, 64, 100)])
def test_stream_compress_exact(sink, recent, length):
    from duo_attn import _hip

    rng = np.random.default_rng(length)
    h = 3
    cap = max(length, sink + recent)
    q, sz = _pools(cap, h)
    vq, vsz = _pools(cap, h)
    for t in (q, vq):
        t.copy_(torch.from_numpy(rng.integers(0, 256, (cap, h, 64), dtype=np.uint8)))
    for t in (sz, vsz):
        t.copy_(torch.from_numpy(rng.standard_normal((cap, h, 2)).astype(np.float16)))
    before = [t.cpu().clone() for t in (q, sz, vq, vsz)]
    n = _hip.int4_stream_compress(q, sz, vq, vsz, length, sink, recent)
    W = sink + recent
    assert n == min(length, W)
    for t, b in zip((q, sz, vq, vsz), before):
        exp = b.clone()
        if length > W:
            exp[sink:W] = b[length - recent:length]
        assert torch.equal(t.cpu()[:n], exp[:n])


def _ref_attention(q, kd, vd, group, budget=None):
    """q [Hq,128] fp32; kd/vd [T,h,128] fp32 (dequantised) -> [Hq,128] fp32, exact softmax.
    ``budget`` (optional [Hq,128] tensor) receives sum_j p_j |v_j|: the scale of the error that rounding P
    to fp16 before P.V (what FA2 does on fp16 inputs, and the MFMA kernel here) may introduce."""
    Hq = q.shape[0]
    out = torch.empty(Hq, 128)
    for hq in range(Hq):
        k, v = kd[:, hq // group], vd[:, hq // group]
        s = (k @ q[hq]) / (128 ** 0.5)
        p = torch.softmax(s, 0)
        out[hq] = p @ v
        if budget is not None:
            budget[hq] = p @ v.abs()
    return out


@gpu
@pytest.mark.parametrize("group,nf,ns,n_full,n_stream", [(4, 1, 1, 1, 1), (4, 2, 6, 300, 385), (4, 8, 0, 5000, 0),
                                                          (4, 0, 8, 0, 384), (1, 4, 4, 777, 100), (2, 3, 1, 40000, 50)])
@pytest.mark.parametrize("odd_rows", [False, True])
@pytest.mark.parametrize("mode", [0, 2])
def test_fused_int4_decode(group, nf, ns, n_full, n_stream, odd_rows, mode):
    """odd_rows: a tenth of the rows are scaled by 1e-3 and a few by several hundred, so their quantisation
    scale leaves [2^-10, 64) and the kernel's per-tile vote takes the subtract-multiply-add dequantisation
    instead of the exact-fma one; both must reproduce the reference's two-rounding values.
    mode 2: the FOLDED kernel (scale / zero applied to the score tile and to P, no per-element dequantisation) — same bar
    against the same reference values; the odd rows there move the V scales' wave-uniform exponent (scales beyond 2^7)."""
    from duo_attn import _hip

    g = torch.Generator().manual_seed(n_full + n_stream)
    Hq = (nf + ns) * group
    q = torch.randn(Hq, 128, generator=g).to(torch.float16)
    ref = torch.empty(Hq, 128)
    bud = torch.empty(Hq, 128)
    pools = []
    for n_h, T, off in ((nf, n_full, 0), (ns, n_stream, nf * group)):
        if n_h == 0:
            pools.append(None)
            continue
        k = torch.randn(T, n_h, 128, generator=g)
        v = torch.randn(T, n_h, 128, generator=g)
        if odd_rows:
            for x, big in ((k, 300.0), (v, 800.0)):
                f = torch.ones(T, n_h, 1)
                u = torch.rand(T, n_h, 1, generator=g)
                f[u < 0.1] = 1e-3
                f[u > 0.995] = big
                x *= f
        k, v = k.to(torch.float16), v.to(torch.float16)
        kq, ksz = _pools(T + 3, n_h)
        vq, vsz = _pools(T + 3, n_h)
        _hip.int4_quantize(k.to(DEV), kq, ksz, 0)
        _hip.int4_quantize(v.to(DEV), vq, vsz, 0)
        pools.append(_hip.make_int4_pool(kq, ksz, vq, vsz, T, off))
        kd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(k.float().numpy())).astype(np.float32))
        vd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(v.float().numpy())).astype(np.float32))
        ref[off:off + n_h * group] = _ref_attention(q.float()[off:off + n_h * group], kd, vd, group,
                                                    bud[off:off + n_h * group])
        pools[-1]._keep = (kq, ksz, vq, vsz)
    out = torch.full((Hq, 128), float("nan"), dtype=torch.float16, device=DEV)
    _hip.attn_decode_int4(q.to(DEV), out, group, pools[0], pools[1], 128 ** -0.5, fused=mode)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    err = (o - ref).abs()
    # 1e-3 relative + one fp16 output ulp + the P-rounding budget (p in fp16: 2^-11 each, 2^-10 allowed)
    tol = 1e-3 * ref.abs() + 2.0 ** -10 * ref.abs() + 2.0 ** -10 * bud + 1e-3 * ref.pow(2).mean().sqrt()
    assert (err <= tol).all(), f"max err {err.max():.3e}"
    from helpers import PARITY_LOG

    rms = ref.pow(2).mean().sqrt()
    PARITY_LOG[f"int4 decode mode={mode} g={group} nf={nf} ns={ns} n={n_full}/{n_stream} odd={odd_rows}"] = {
        "n": int(o.numel()), "max_abs_err": float(err.max()), "rms_err": float((o - ref).pow(2).mean().sqrt()),
        "rms_ref": float(rms), "rms_err_over_rms_ref": float((o - ref).pow(2).mean().sqrt() / rms),
        "worst_err_over_elementwise_tol": float((err / tol.clamp_min(1e-30)).max())}


@gpu
def test_int4_cache_put_compress_decode_flow():
    """DuoAttentionStaticINT4KVCache: chunked put + compress, then decode steps — against the reference's
    control flow restated with the oracle (demo/int4_kv.py:261-492, demo/w8a8kv4_llama.py:219-278)."""
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache

    counts, Hq, Hkv, sink, recent, chunk = [1, 3, 0, 4], 16, 4, 16, 48, 200
    L = len(counts)
    model = ShapeModel(L, Hq, Hkv, 128, device=DEV, dtype=torch.float16)
    cache = DuoAttentionStaticINT4KVCache(model, heads_from_counts(counts, Hkv), 1, 700, sink, recent, chunk)
    g = torch.Generator().manual_seed(7)
    W = sink + recent
    ref_full = [[np.zeros((0, nf, 128), np.float16)] * 2 for nf in counts]
    ref_str = [[np.zeros((0, Hkv - nf, 128), np.float16)] * 2 for nf in counts]

    def dq(x):
        return dequantize_int4_ref(*quantize_int4_ref(x.astype(np.float32)))

    for S in (200, 150, 1, 1, 1):
        for l, nf in enumerate(counts):
            k = torch.randn(1, S, Hkv, 128, generator=g).to(torch.float16)
            v = torch.randn(1, S, Hkv, 128, generator=g).to(torch.float16)
            q = torch.randn(1, S, Hq, 128, generator=g).to(torch.float16)
            fk, fv, sk, sv = cache.put(l, k.to(DEV), v.to(DEV))
            kn, vn = k[0].numpy(), v[0].numpy()
            ref_full[l] = [np.concatenate([ref_full[l][0], dq(kn[:, :nf])]), np.concatenate([ref_full[l][1], dq(vn[:, :nf])])]
            ref_str[l] = [np.concatenate([ref_str[l][0], dq(kn[:, nf:])]), np.concatenate([ref_str[l][1], dq(vn[:, nf:])])]
            if nf:
                assert np.array_equal(fk.cpu().numpy()[0].view(np.uint16), ref_full[l][0].view(np.uint16))
                assert np.array_equal(fv.cpu().numpy()[0].view(np.uint16), ref_full[l][1].view(np.uint16))
            if Hkv - nf:
                assert np.array_equal(sk.cpu().numpy()[0].view(np.uint16), ref_str[l][0].view(np.uint16))
            if S == 1:
                out = cache.decode_attention(l, q.to(DEV)).float().cpu()[0, 0]
                G = Hq // Hkv
                ref = torch.empty(Hq, 128)
                if nf:
                    ref[:nf * G] = _ref_attention(q[0, 0, :nf * G].float(), torch.from_numpy(ref_full[l][0].astype(np.float32)),
                                                  torch.from_numpy(ref_full[l][1].astype(np.float32)), G)
                if Hkv - nf:
                    ref[nf * G:] = _ref_attention(q[0, 0, nf * G:].float(), torch.from_numpy(ref_str[l][0].astype(np.float32)),
                                                  torch.from_numpy(ref_str[l][1].astype(np.float32)), G)
                err = (out - ref).abs()
                tol = 1e-3 * ref.abs() + 2.0 ** -10 * ref.abs() + 1e-3 * ref.pow(2).mean().sqrt()
                assert (err <= tol).all(), (S, l, float(err.max()))
            cache.compress(l)
            if Hkv - nf and ref_str[l][0].shape[0] > W:
                ref_str[l] = [np.concatenate([a[:sink], a[-recent:]]) for a in ref_str[l]]
            assert cache.kv_seq_len_list[l] == ref_full[l][0].shape[0]
            if Hkv - nf:
                assert cache.streaming_kv_seq_len_list[l] == ref_str[l][0].shape[0]
            else:
                ref_str[l] = [a[:0] for a in ref_str[l]]
    assert cache.memory_usage == sum(2 * 68 * (nf * 700 + (Hkv - nf) * (W + chunk)) for nf in counts)


@gpu
@pytest.mark.parametrize("S,group,nf,ns,la,ls", [(2, 4, 1, 1, 3, 2), (100, 4, 1, 3, 1000, 384), (257, 4, 2, 0, 300, 0),
                                                  (300, 4, 0, 2, 0, 384), (513, 1, 2, 2, 77, 10), (700, 8, 1, 0, 129, 0),
                                                  (1024, 2, 1, 1, 1100, 384)])
@pytest.mark.parametrize("flags", [0, 128], ids=["w4x64", "w8x32"])
def test_fp16_prefill_kernel(S, group, nf, ns, la, ls, flags):
    """duo_attn_prefill_f16 (the MFMA prefill kernel instantiated for fp16) against the oracle, both head
    classes, pool + chunk as two segments."""
    from duo_attn.backend import HipBackend
    from helpers import attn_close
    from oracle.duo_oracle import flash_attn_func_ref

    g = torch.Generator().manual_seed(S + la)
    Hq = (nf + ns) * group
    r = lambda *shape: torch.randn(*shape, generator=g).to(torch.float16)
    q, kn, vn = r(S, Hq, 128), r(S, nf + ns, 128), r(S, nf + ns, 128)
    out = torch.full((S, Hq, 128), float("nan"), dtype=torch.float16, device=DEV)
    ref, bud = torch.empty(S, Hq, 128), torch.empty(S, Hq, 128)
    kw = dict(round_p=False, out_dtype=torch.float32, return_budget=True)
    full = stream = None
    knd, vnd = kn.to(DEV), vn.to(DEV)
    if nf:
        fk, fv = r(la, nf, 128), r(la, nf, 128)
        full = (nf, 0, (fk.to(DEV), fv.to(DEV)) if la else None, (knd[:, :nf], vnd[:, :nf]))
        o, b = flash_attn_func_ref(q[None, :, :nf * group], torch.cat([fk, kn[:, :nf]])[None], torch.cat([fv, vn[:, :nf]])[None], **kw)
        ref[:, :nf * group], bud[:, :nf * group] = o[0], b[0]
    if ns:
        sk, sv = r(ls, ns, 128), r(ls, ns, 128)
        stream = (ns, nf * group, (sk.to(DEV), sv.to(DEV)) if ls else None, (knd[:, nf:], vnd[:, nf:]))
        o, b = flash_attn_func_ref(q[None, :, nf * group:], torch.cat([sk, kn[:, nf:]])[None], torch.cat([sv, vn[:, nf:]])[None], **kw)
        ref[:, nf * group:], bud[:, nf * group:] = o[0], b[0]
    from duo_attn import _hip
    _hip.set_debug_flags(flags)       # bit 7: the 8-wave kernel; default: the fp16 twin of the 4-wave x 64-row kernel
    try:
        HipBackend().attention(q.to(DEV), out, group, full, stream, 128 ** -0.5)
        torch.cuda.synchronize()
    finally:
        _hip.set_debug_flags(0)
    attn_close(out, ref, f"fp16 prefill S={S} flags={flags}", bud)


@gpu
def test_int4_cache_chunked_prefill_attention():
    """DuoAttentionStaticINT4KVCache.prefill_attention: first chunk over the raw chunk, later chunks over the
    dequantised pools (incl. the chunk's own quantised rows), against the oracle on oracle-dequantised data
    (reference demo/w8a8kv4_llama.py:219-278)."""
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache
    from helpers import attn_close
    from oracle.duo_oracle import flash_attn_func_ref

    counts, Hq, Hkv, sink, recent, chunk = [1, 3, 0, 4], 16, 4, 16, 48, 300
    G = Hq // Hkv
    model = ShapeModel(len(counts), Hq, Hkv, 128, device=DEV, dtype=torch.float16)
    cache = DuoAttentionStaticINT4KVCache(model, heads_from_counts(counts, Hkv), 1, 900, sink, recent, chunk)
    g = torch.Generator().manual_seed(5)
    W = sink + recent
    dq = lambda x: torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(x.float().numpy())).astype(np.float32))
    hist = [dict(fk=torch.zeros(0, nf, 128), fv=torch.zeros(0, nf, 128), sk=torch.zeros(0, Hkv - nf, 128),
                 sv=torch.zeros(0, Hkv - nf, 128)) for nf in counts]
    kw = dict(round_p=False, out_dtype=torch.float32, return_budget=True)
    for ci, S in enumerate((300, 257, 64)):
        for l, nf in enumerate(counts):
            q = torch.randn(1, S, Hq, 128, generator=g).to(torch.float16)
            k = torch.randn(1, S, Hkv, 128, generator=g).to(torch.float16)
            v = torch.randn(1, S, Hkv, 128, generator=g).to(torch.float16)
            cache.put(l, k.to(DEV), v.to(DEV), dequantize=False)
            out = cache.prefill_attention(l, q.to(DEV), k.to(DEV), v.to(DEV))
            h = hist[l]
            h["fk"], h["fv"] = torch.cat([h["fk"], dq(k[0, :, :nf])]), torch.cat([h["fv"], dq(v[0, :, :nf])])
            h["sk"], h["sv"] = torch.cat([h["sk"], dq(k[0, :, nf:])]), torch.cat([h["sv"], dq(v[0, :, nf:])])
            ref, bud = torch.empty(S, Hq, 128), torch.empty(S, Hq, 128)
            if ci == 0:
                o, b = flash_attn_func_ref(q, k, v, **kw)
                ref, bud = o[0], b[0]
            else:
                if nf:
                    o, b = flash_attn_func_ref(q[:, :, :nf * G], h["fk"][None], h["fv"][None], **kw)
                    ref[:, :nf * G], bud[:, :nf * G] = o[0], b[0]
                if Hkv - nf:
                    o, b = flash_attn_func_ref(q[:, :, nf * G:], h["sk"][None], h["sv"][None], **kw)
                    ref[:, nf * G:], bud[:, nf * G:] = o[0], b[0]
            attn_close(out[0], ref, f"int4 chunk {ci} layer {l}", bud)
            cache.compress(l)
            if h["sk"].shape[0] > W:        # the reference keeps sink + recent rows of the streaming pool
                h["sk"] = torch.cat([h["sk"][:sink], h["sk"][-recent:]])
                h["sv"] = torch.cat([h["sv"][:sink], h["sv"][-recent:]])


@gpu
def test_int4_cache_batch_rows_equal_single_rows():
    """B = 2 through DuoAttentionStaticINT4KVCache (put / chunked prefill attention in ONE batched fp16 launch / decode /
    compress) == each row through its own B = 1 cache: packed pools bit for bit, prefill outputs bit for bit with the
    key-range split disabled (same tiles), decode outputs bit for bit (ONE batched launch pair for both rows — the batch row
    is grid.z of the same kernel, every row with its own partial area — against one launch pair per row)."""
    from duo_attn import _hip
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache

    counts, Hq, Hkv, sink, recent, chunk = [1, 3, 0, 4], 16, 4, 16, 48, 300
    heads = heads_from_counts(counts, Hkv)
    model = ShapeModel(len(counts), Hq, Hkv, 128, device=DEV, dtype=torch.float16)
    both = DuoAttentionStaticINT4KVCache(model, heads, 2, 900, sink, recent, chunk)
    solo = [DuoAttentionStaticINT4KVCache(model, heads, 1, 900, sink, recent, chunk) for _ in range(2)]
    g = torch.Generator().manual_seed(9)
    _hip.set_debug_flags(256)
    try:
        for S in (300, 257, 1, 1):
            for l in range(len(counts)):
                q = torch.randn(2, S, Hq, 128, generator=g).to(torch.float16).to(DEV)
                k = torch.randn(2, S, Hkv, 128, generator=g).to(torch.float16).to(DEV)
                v = torch.randn(2, S, Hkv, 128, generator=g).to(torch.float16).to(DEV)
                outs = []
                for cache, sl in [(both, slice(0, 2))] + [(solo[b], slice(b, b + 1)) for b in range(2)]:
                    past = cache.kv_seq_len_list[l]
                    cache.put(l, k[sl], v[sl], dequantize=False)
                    o = cache.decode_attention(l, q[sl]) if (S == 1 and past > 0) else cache.prefill_attention(l, q[sl], k[sl], v[sl])
                    cache.compress(l)
                    outs.append(o)
                assert torch.equal(outs[0][0:1], outs[1]) and torch.equal(outs[0][1:2], outs[2]), (S, l)
    finally:
        _hip.set_debug_flags(0)
    for l in range(len(counts)):
        n, m = both.kv_seq_len_list[l], both.streaming_kv_seq_len_list[l]
        for b in range(2):
            assert (solo[b].kv_seq_len_list[l], solo[b].streaming_kv_seq_len_list[l]) == (n, m)
            for name, rows in (("full_key_caches", n), ("full_value_caches", n), ("streaming_key_caches", m), ("streaming_value_caches", m)):
                A, Bc = getattr(both, name)[l], getattr(solo[b], name)[l]
                assert torch.equal(A.quantized_data[b, :rows], Bc.quantized_data[0, :rows]), (name, l, b)
                assert torch.equal(A.scale_zero[b, :rows], Bc.scale_zero[0, :rows]), (name, l, b)


@gpu
@pytest.mark.parametrize("case", ["tiny_scales", "huge_scales", "constant_rows", "mixed", "token_major"])
def test_folded_int4_decode_scale_extremes(case):
    """the folded kernel keeps P' = p s' 2^-E inside fp16 for ANY scales (wave-uniform exponent E, re-centred by the tile that
    leaves the window): pools whose V scales are all tiny / all huge / zero (constant rows) / wildly mixed still meet the bar;
    token-major pools (token_stride_rows != 1) run through the dequantising kernel and agree as well"""
    from duo_attn import _hip

    g = torch.Generator().manual_seed(len(case))
    group, n_h, T = 4, 2, 3000
    q = torch.randn(n_h * group, 128, generator=g).to(torch.float16)
    k = torch.randn(T, n_h, 128, generator=g)
    v = torch.randn(T, n_h, 128, generator=g)
    if case == "tiny_scales":
        v *= 3e-4                                  # V scales ~1e-4 < 2^-12: outside the folded window from below
    elif case == "huge_scales":
        v *= 2000.0
    elif case == "constant_rows":
        v[:] = v[:, :, :1]                       # every row constant: scale = 1e-8 -> 0 in fp16, values = zero point
        v[::7] = torch.randn(len(v[::7]), n_h, 128, generator=g)
    elif case == "mixed":
        f = torch.ones(T, n_h, 1)
        u = torch.rand(T, n_h, 1, generator=g)
        f[u < 0.3] = 1e-4
        f[u > 0.9] = 3000.0
        v *= f
        k *= 1.0 + 50.0 * (u > 0.97)
    k, v = k.to(torch.float16), v.to(torch.float16)
    if case == "token_major":
        kq = torch.zeros(T + 3, n_h, 64, dtype=torch.uint8, device=DEV)
        ksz = torch.zeros(T + 3, n_h, 2, dtype=torch.float16, device=DEV)
        vq, vsz = torch.zeros_like(kq), torch.zeros_like(ksz)
    else:
        kq, ksz = _pools(T + 3, n_h)
        vq, vsz = _pools(T + 3, n_h)
    _hip.int4_quantize(k.to(DEV), kq, ksz, 0)
    _hip.int4_quantize(v.to(DEV), vq, vsz, 0)
    pool = _hip.make_int4_pool(kq, ksz, vq, vsz, T, 0)
    kd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(k.float().numpy())).astype(np.float32))
    vd = torch.from_numpy(dequantize_int4_ref(*quantize_int4_ref(v.float().numpy())).astype(np.float32))
    bud = torch.empty(n_h * group, 128)
    ref = _ref_attention(q.float(), kd, vd, group, bud)
    outs = {}
    for mode in (0, 2):
        out = torch.full((n_h * group, 128), float("nan"), dtype=torch.float16, device=DEV)
        _hip.attn_decode_int4(q.to(DEV), out, group, pool, None, 128 ** -0.5, fused=mode)
        o = out.float().cpu()
        assert torch.isfinite(o).all(), (case, mode)
        err = (o - ref).abs()
        # (+ one fp16 denormal step: the tiny-scale case's outputs sit at the bottom of the fp16 range)
        tol = 1e-3 * ref.abs() + 2.0 ** -10 * ref.abs() + 2.0 ** -10 * bud + 1e-3 * ref.pow(2).mean().sqrt() + 2.0 ** -24
        assert (err <= tol).all(), f"{case} mode {mode}: max err {err.max():.3e} (ref rms {ref.pow(2).mean().sqrt():.3e})"
        outs[mode] = o
    if case == "token_major":
        assert torch.equal(outs[0], outs[2])        # the folded form needs head-major pools: this launch ran as mode 0


@gpu
def test_folded_decode_is_opt_in_and_within_the_value_rounding_budget():
    """DuoAttentionStaticINT4KVCache(folded_decode=True): same pools, decode outputs within one fp16 ulp per dequantised value
    (2^-10 sum p |v|) of the default (dequantising) decode; the default is the dequantising kernel"""
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache

    counts, Hq, Hkv = [1, 3, 0, 4], 16, 4
    model = ShapeModel(len(counts), Hq, Hkv, 128, device=DEV, dtype=torch.float16)
    mk = lambda **kw: DuoAttentionStaticINT4KVCache(model, heads_from_counts(counts, Hkv), 1, 700, 16, 48, 200, **kw)
    a, b = mk(), mk(folded_decode=True)
    assert a.folded_decode is False and b.folded_decode is True
    g = torch.Generator().manual_seed(3)
    for S in (200, 150, 1, 1):
        for l in range(len(counts)):
            k = torch.randn(1, S, Hkv, 128, generator=g).to(torch.float16).to(DEV)
            v = torch.randn(1, S, Hkv, 128, generator=g).to(torch.float16).to(DEV)
            q = torch.randn(1, S, Hq, 128, generator=g).to(torch.float16).to(DEV)
            for c in (a, b):
                c.put(l, k, v, dequantize=False)
            if S == 1:
                oa, ob = a.decode_attention(l, q).float(), b.decode_attention(l, q).float()
                _, fv, _, sv = a.get(l)
                vmax = max(float(t.abs().max()) for t in (fv, sv) if t.numel())
                err = (oa - ob).abs()
                assert (err <= 2.0 ** -10 * vmax + 2.0 ** -10 * oa.abs() + 1e-3 * oa.pow(2).mean().sqrt()).all(), float(err.max())
                assert not torch.equal(oa, ob)           # (it really is the other kernel)
            for c in (a, b):
                c.compress(l)
