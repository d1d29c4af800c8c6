"""GPU parity at the FULL sizes of BASELINE.json's configs.

The CPU oracle cannot finish these sizes, so each case is checked two ways:
  * sampled parity — a few dozen query rows (all q heads) against an exact fp32 softmax attention
    written in plain torch on the GPU from the same inputs (the visible key set per row built directly
    from the SURVEY §8(a7) formula), with the helpers.attn_close tolerance;
  * a size-independent property — with every V row of a head equal to one vector, the output must be
    that vector (softmax weights sum to one whatever the 10^5..10^6 keys are), to one output ulp.
cfg1 Llama-2-7B shape (MHA, linear rope factor 8), 4K;  cfg2 Llama-3-8B shape, 128K, chunk 16384;
cfg3 Mistral shape, 32K, chunk 32000;  cfg4 the single-GPU share of the 1M-token prefill (chunk 32000
at past ~1M);  cfg5 int4 pools, 3.3M-token decode.
"""
import numpy as np
import pytest
import torch

from helpers import attn_close, with_rounded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _be():
    from duo_attn.backend import HipBackend

    return HipBackend()


def _randn(shape, g, dtype=torch.bfloat16):
    return torch.randn(shape, generator=g, device=DEV, dtype=torch.float32).to(dtype)


def _pool(n_heads, rows, g, dtype=torch.bfloat16):
    """head-major storage, token-major view [rows, n_heads, D] (as DuoAttentionStaticKVCache allocates)"""
    return _randn((n_heads, rows, D), g, dtype).permute(1, 0, 2)


def _ref_rows(q_rows, K, V, vis, scale):
    """q_rows [n, G, D], K/V [T, D] (one kv head), vis [n] = number of visible keys (prefix) per row.
    Returns (out [n, G, D] fp32, budget [n, G, D] = sum_j p_j |v_j|, the same rows in the REFERENCE's arithmetic — FA2:
    exp(s - max) rounded to the element type before P.V, normalised by the unrounded sum, output rounded — for
    helpers.with_rounded)."""
    s = torch.einsum("ngd,td->ngt", q_rows.float(), K.float()) * scale
    t = torch.arange(K.shape[0], device=K.device)
    s.masked_fill_(t[None, None, :] >= vis[:, None, None], float("-inf"))
    p = torch.softmax(s, dim=-1)
    e = torch.exp(s - s.amax(dim=-1, keepdim=True))
    rounded = (torch.einsum("ngt,td->ngd", e.to(K.dtype).float(), V.float()) / e.sum(dim=-1, keepdim=True)).to(K.dtype).float()
    return torch.einsum("ngt,td->ngd", p, V.float()), torch.einsum("ngt,td->ngd", p, V.float().abs()), rounded


def _check_prefill(S, past, nf, ns, G, W, g, n_sample=48, first_chunk=False):
    """one later-chunk (or first-chunk) prefill launch at full size, sampled parity for both head classes"""
    be = _be()
    nkv, Hq, scale = nf + ns, (nf + ns) * G, D ** -0.5
    q = _randn((S, Hq, D), g)
    kn, vn = _pool(nkv, S, g), _pool(nkv, S, g)
    out = torch.empty_like(q)
    rows = torch.randint(0, S, (n_sample,), generator=g, device=DEV)
    rows[0], rows[1] = 0, S - 1
    ref = torch.empty(n_sample, Hq, D, device=DEV)
    bud = torch.empty_like(ref)
    rnd = torch.empty_like(ref)
    if first_chunk:
        be.attention(q, out, G, (nkv, 0, None, (kn, vn)), None, scale)
        for h in range(nkv):
            r, b, rr = _ref_rows(q[rows, h * G:(h + 1) * G], kn[:, h], vn[:, h], rows + 1, scale)
            ref[:, h * G:(h + 1) * G], bud[:, h * G:(h + 1) * G], rnd[:, h * G:(h + 1) * G] = r, b, rr
    else:
        fk, fv = _pool(max(nf, 1), past + S, g), _pool(max(nf, 1), past + S, g)
        sk, sv = _pool(max(ns, 1), W, g), _pool(max(ns, 1), W, g)
        fk[past:, :nf] = kn[:, :nf]      # put_full_kv already happened: the new rows are in the pool
        fv[past:, :nf] = vn[:, :nf]
        full = (nf, 0, (fk[:past, :nf], fv[:past, :nf]), (fk[past:, :nf], fv[past:, :nf])) if nf else None
        stream = (ns, nf * G, (sk[:, :ns], sv[:, :ns]), (kn[:, nf:], vn[:, nf:])) if ns else None
        be.attention(q, out, G, full, stream, scale)
        for h in range(nf):          # retrieval head: keys {0 .. past + row}
            r, b, rr = _ref_rows(q[rows, h * G:(h + 1) * G], fk[:, h], fv[:, h], past + rows + 1, scale)
            ref[:, h * G:(h + 1) * G], bud[:, h * G:(h + 1) * G], rnd[:, h * G:(h + 1) * G] = r, b, rr
        for j in range(ns):          # streaming head: Pool(past) U {past .. past + row}
            h = nf + j
            K = torch.cat([sk[:, j], kn[:, h]], 0)
            V = torch.cat([sv[:, j], vn[:, h]], 0)
            r, b, rr = _ref_rows(q[rows, h * G:(h + 1) * G], K, V, W + rows + 1, scale)
            ref[:, h * G:(h + 1) * G], bud[:, h * G:(h + 1) * G], rnd[:, h * G:(h + 1) * G] = r, b, rr
    torch.cuda.synchronize()
    attn_close(out[rows], ref, f"prefill S={S} past={past} nf={nf} ns={ns}", with_rounded(bud, rnd))
    return q, out


@pytest.mark.parametrize("flags", [0, 128], ids=["w4x64", "w8x32"])
def test_cfg2_llama3_128k_last_chunk(flags):
    """(both prefill kernels at the bench workload's own launch shape)"""
    from duo_attn import _hip

    g = torch.Generator(device=DEV).manual_seed(2)
    _hip.set_debug_flags(flags)
    try:
        _check_prefill(S=16384, past=131072 - 16384, nf=4, ns=4, G=4, W=384, g=g)
    finally:
        _hip.set_debug_flags(0)


def test_cfg2_llama3_128k_first_chunk():
    g = torch.Generator(device=DEV).manual_seed(3)
    _check_prefill(S=16384, past=0, nf=8, ns=0, G=4, W=384, g=g, first_chunk=True)


def test_cfg3_mistral_32k_chunk_32000():
    """32768 tokens in chunks of 32000 (scripts/efficiency.sh default): 32000 (first chunk) + 768 at past
    32000 — odd chunk sizes, neither a multiple of the 256-row query tile nor of the 64-key tile."""
    g = torch.Generator(device=DEV).manual_seed(4)
    _check_prefill(S=32000, past=0, nf=8, ns=0, G=4, W=384, g=g, first_chunk=True)
    _check_prefill(S=768, past=32000, nf=3, ns=5, G=4, W=384, g=g)


def test_cfg4_1m_context_chunk_32000():
    """the work one pipeline stage does per layer near the end of the 1M-token prefill (cfg4)"""
    g = torch.Generator(device=DEV).manual_seed(5)
    _check_prefill(S=32000, past=1_048_576 - 32000, nf=2, ns=6, G=4, W=384, g=g, n_sample=24)


def test_cfg1_llama2_shape_4k():
    """MHA (group 1), 32 kv heads: 4K first chunk, then one later chunk with 8 retrieval / 24 streaming"""
    g = torch.Generator(device=DEV).manual_seed(6)
    _check_prefill(S=4096, past=0, nf=32, ns=0, G=1, W=320, g=g, first_chunk=True)
    _check_prefill(S=1024, past=4096, nf=8, ns=24, G=1, W=320, g=g)


@pytest.mark.parametrize("N,nf,ns,G", [(131072, 4, 4, 4), (131072, 8, 0, 4), (1_048_576, 2, 6, 4), (4096, 8, 24, 1)])
def test_decode_step_full_size(N, nf, ns, G):
    """duo_decode_layer_bf16 (fused step) with N cached tokens: every q head against the exact fp32
    attention over the rotated inputs; RoPE done here in fp32 torch with the same angle arithmetic."""
    from duo_attn import _hip

    g = torch.Generator(device=DEV).manual_seed(N % 1000 + nf)
    nkv, Hq, W, sink, recent = nf + ns, (nf + ns) * G, 384, 128, 256
    theta, rscale, pos = 3580165449.0, 1.0, N
    q, k, v = _randn((Hq, D), g), _randn((nkv, D), g), _randn((nkv, D), g)
    fk, fv = _pool(max(nf, 1), N + 2, g)[:, :nf], _pool(max(nf, 1), N + 2, g)[:, :nf]
    sk, sv = _pool(max(ns, 1), W, g)[:, :ns], _pool(max(ns, 1), W, g)[:, :ns]
    sk0, sv0 = sk.clone(), sv.clone()
    out = torch.empty_like(q)
    n = _hip.decode_layer(q, k, v, out, nf, fk, fv, N, sk, sv, W, sink, recent, pos, rscale, theta, D ** -0.5)
    assert n == W

    inv = torch.tensor([float(np.float32(theta ** (-2.0 * i / D) / rscale)) for i in range(D // 2)], device=DEV)
    ang = torch.tensor(float(pos), device=DEV) * inv
    cos, sin = torch.cos(ang), torch.sin(ang)

    def rope(x):
        xf = x.float()
        lo, hi = xf[..., :D // 2], xf[..., D // 2:]
        return torch.cat([lo * cos - hi * sin, hi * cos + lo * sin], -1).to(torch.bfloat16)

    qr, kr = rope(q), rope(k)
    ref = torch.empty(Hq, D, device=DEV)
    one = torch.ones(1, dtype=torch.long, device=DEV)
    for h in range(nf):
        K, V = torch.cat([fk[:N, h], kr[h:h + 1]], 0), torch.cat([fv[:N, h], v[h:h + 1]], 0)
        ref[h * G:(h + 1) * G] = _ref_rows(qr[None, h * G:(h + 1) * G], K, V, one * (N + 1), D ** -0.5)[0][0]
    for j in range(ns):
        h = nf + j
        K, V = torch.cat([sk0[:, j], kr[h:h + 1]], 0), torch.cat([sv0[:, j], v[h:h + 1]], 0)
        ref[h * G:(h + 1) * G] = _ref_rows(qr[None, h * G:(h + 1) * G], K, V, one * (W + 1), D ** -0.5)[0][0]
    attn_close(out, ref, f"decode N={N} nf={nf} ns={ns}")
    # the pools after the step: rotated key + value appended at row N; streaming pool = sink ++ last `recent`
    if nf:
        assert torch.equal(fv[N], v[:nf])
        assert ((fk[N].float() - kr[:nf].float()).abs() <= kr[:nf].float().abs() * 2.0 ** -7 + 1e-6).all()
    if ns:
        assert torch.equal(sv[:sink], sv0[:sink]) and torch.equal(sv[sink:W - 1], sv0[sink + 1:]) and torch.equal(sv[W - 1], v[nf:])
        assert torch.equal(sk[:sink], sk0[:sink]) and torch.equal(sk[sink:W - 1], sk0[sink + 1:])


def test_constant_value_rows_give_that_vector_at_128k():
    """softmax weights sum to one: with V[j] = c for every key of a head, the output is c — decode over
    131072 keys, and prefill rows attending to 131072 keys."""
    be = _be()
    g = torch.Generator(device=DEV).manual_seed(9)
    G, nf, ns, N, S, W = 4, 2, 2, 131072, 512, 384
    Hq, scale = (nf + ns) * G, D ** -0.5
    c = _randn((nf + ns, D), g)
    fk = _pool(nf, N + S, g)
    const = lambda cv, rows: cv[:, None, :].expand(cv.shape[0], rows, D).contiguous().permute(1, 0, 2)   # head-major like K
    fv = const(c[:nf], N + S)
    sk = _pool(ns, W, g)
    sv = const(c[nf:], W)
    for S_ in (1, S):
        q = _randn((S_, Hq, D), g)
        kn = _pool(nf + ns, S_, g)
        vn = const(c, S_)
        out = torch.empty_like(q)
        full = (nf, 0, (fk[:N], fv[:N]), (kn[:, :nf], vn[:, :nf]))
        stream = (ns, nf * G, (sk, sv), (kn[:, nf:], vn[:, nf:]))
        be.attention(q, out, G, full, stream, scale)
        want = c.float().repeat_interleave(G, 0)[None].expand(S_, Hq, D)
        err = (out.float() - want).abs()
        # fp32 accumulation of ~1e5 weights that sum to one, then one bf16 rounding of the output
        assert (err <= want.abs() * 2.0 ** -7 + 1e-5).all(), f"S={S_}: max err {err.max():.3e}"


@pytest.mark.parametrize("mode", [0, 2])
def test_cfg5_int4_decode_3m_tokens(mode):
    """int4 pools, 3.3M cached tokens, one layer with 2 retrieval + 2 streaming kv heads: against exact fp32
    attention over pools dequantised by the oracle (oracle/int4_oracle.py, pinned to the reference's own
    kernel by tests/test_int4_golden.py; its torch form runs on the device, and a sample of rows is
    re-checked here against the numpy form)."""
    import numpy as np

    from duo_attn import _hip
    from oracle.int4_oracle import dequantize_int4_ref, dequantize_int4_torch

    g = torch.Generator(device=DEV).manual_seed(11)
    G, nf, ns, N, W = 4, 2, 2, 3_300_000, 384
    Hq, scale = (nf + ns) * G, D ** -0.5

    def pools(h, T):
        q = torch.randint(0, 256, (h, T, 64), generator=g, device=DEV, dtype=torch.uint8).permute(1, 0, 2)
        sz = torch.empty(h, T, 2, device=DEV, dtype=torch.float16)
        sz[..., 0] = (torch.rand(h, T, generator=g, device=DEV) * 0.3 + 0.002).to(torch.float16)
        sz[..., 1] = (torch.randn(h, T, generator=g, device=DEV)).to(torch.float16)
        return q, sz.permute(1, 0, 2)

    def dequant(qp, sz):     # [T, 64] u8, [T, 2] f16 -> [T, 128] fp16 values as fp32: the oracle's function
        return dequantize_int4_torch(qp, sz).float()

    q = _randn((Hq, D), g, torch.float16)
    out = torch.empty_like(q)
    fkq, fksz = pools(nf, N)
    fvq, fvsz = pools(nf, N)
    skq, sksz = pools(ns, W)
    svq, svsz = pools(ns, W)
    full = _hip.make_int4_pool(fkq, fksz, fvq, fvsz, N, 0)
    stream = _hip.make_int4_pool(skq, sksz, svq, svsz, W, nf * G)
    _hip.attn_decode_int4(q, out, G, full, stream, scale, fused=mode)       # (mode 2: the folded kernel)
    rows = torch.randint(0, N, (8192,), generator=g, device=DEV)
    sp, ssz = fkq[rows, 0].cpu(), fksz[rows, 0].cpu()
    assert np.array_equal(dequantize_int4_torch(fkq[rows, 0], fksz[rows, 0]).cpu().numpy().view(np.uint16),
                          dequantize_int4_ref(sp.numpy(), ssz[:, 0].numpy(), ssz[:, 1].numpy()).view(np.uint16))
    ref = torch.empty(Hq, D, device=DEV)
    bud = torch.empty_like(ref)
    one = torch.ones(1, dtype=torch.long, device=DEV)
    for cls_off, (kq_, ksz_, vq_, vsz_, n_h, T) in ((0, (fkq, fksz, fvq, fvsz, nf, N)), (nf, (skq, sksz, svq, svsz, ns, W))):
        for j in range(n_h):
            h = cls_off + j
            K, V = dequant(kq_[:, j], ksz_[:, j]), dequant(vq_[:, j], vsz_[:, j])
            r, b, _ = _ref_rows(q[None, h * G:(h + 1) * G], K, V, one * T, scale)
            ref[h * G:(h + 1) * G], bud[h * G:(h + 1) * G] = r[0], b[0]
            del K, V
    o, r = out.float(), ref
    err = (o - r).abs()
    tol = 1e-3 * r.abs() + 2.0 ** -10 * r.abs() + 2.0 ** -10 * bud + 1e-3 * r.pow(2).mean().sqrt()
    assert torch.isfinite(o).all() and (err <= tol).all(), f"max err {err.max():.3e}"


# ----------------------------------------------------------------------------------------------------------------------
# size-independent properties at the bench workload's own size (131072 cached rows): identities the attention obeys whatever
# the data, checked between two runs of the SAME kernel — no reference needed, so they hold at sizes no oracle finishes
# ----------------------------------------------------------------------------------------------------------------------
def _segments(S, g, nf=2, ns=2, G=4, N=131072, W=384):
    """one later-chunk call (S new rows on N cached ones): q, the two head classes' segments, as the static forward builds them"""
    q = _randn((S, (nf + ns) * G, D), g)
    fk, fv = _pool(nf, N + S, g), _pool(nf, N + S, g)
    sk, sv = _pool(ns, W, g), _pool(ns, W, g)
    kn, vn = _pool(ns, S, g), _pool(ns, S, g)

    def call(fv_=fv, sv_=sv, vn_=vn, fk_=fk, q_=q):
        out = torch.empty_like(q_)
        full = (nf, 0, (fk_[:N], fv_[:N]), (fk_[N:], fv_[N:]))
        stream = (ns, nf * G, (sk, sv_), (kn, vn_))
        _be().attention(q_, out, G, full, stream, D ** -0.5)
        torch.cuda.synchronize()
        return out

    return call, dict(q=q, fk=fk, fv=fv, sk=sk, sv=sv, kn=kn, vn=vn, nf=nf, ns=ns, G=G, N=N)


@pytest.mark.parametrize("S", [1, 768], ids=["decode", "prefill"])
def test_value_scaling_by_powers_of_two_is_exact_at_128k(S):
    """attention is linear in V, and a power-of-two factor is exact in bf16 and in the fp32 accumulators: out(4 V) == 4 out(V)
    BIT FOR BIT — split-KV decode over 131072 rows (partials and their merge included) and MFMA prefill rows attending to
    131072 + 768 keys (bf16 P included: P does not depend on V).  The decode's scalar fp32 FMAs are sign-symmetric as well
    (out(-V) == -out(V)); the matrix cores are NOT — measured here: negating V moves single output bits of the MFMA path
    (the accumulation inside v_mfma does not round sign-symmetrically) — so the prefill case is held to exponent shifts."""
    g = torch.Generator(device=DEV).manual_seed(21 + S)
    call, t = _segments(S, g)
    base = call()
    assert torch.isfinite(base).all() and base.float().abs().max() < 1e3
    for f in ((4.0, -1.0, 0.125) if S == 1 else (4.0, 0.125, 2.0 ** -6)):
        got = call(fv_=t["fv"] * f, sv_=t["sv"] * f, vn_=t["vn"] * f)
        assert torch.equal(got, base * f), f"factor {f}: {(got.float() - base.float() * f).abs().max():.3e}"


@pytest.mark.parametrize("S", [1, 768], ids=["decode", "prefill"])
def test_heads_do_not_see_each_other_at_128k(S):
    """a kv head's q heads depend on that head's K / V only: replacing the K and V rows of ONE retrieval head and ONE
    streaming head leaves every other q head's output bit-identical (no partial, tile or workspace slot is shared across
    heads), and changes the replaced heads' own outputs"""
    g = torch.Generator(device=DEV).manual_seed(23 + S)
    call, t = _segments(S, g)
    base = call()
    G, nf = t["G"], t["nf"]
    fk2, fv2, sv2 = t["fk"].clone(), t["fv"].clone(), t["sv"].clone()
    fk2[:, 1] = _randn(fk2[:, 1].shape, g)
    fv2[:, 1] = _randn(fv2[:, 1].shape, g)
    sv2[:, 0] = _randn(sv2[:, 0].shape, g)
    got = call(fk_=fk2, fv_=fv2, sv_=sv2)
    touched = list(range(1 * G, 2 * G)) + list(range(nf * G, (nf + 1) * G))
    same = [h for h in range(base.shape[1]) if h not in touched]
    assert torch.equal(got[:, same], base[:, same])
    assert not torch.equal(got[:, touched], base[:, touched])


def test_cached_rows_of_a_retrieval_head_may_come_in_any_order_at_128k():
    """softmax attention over a SET of keys: a decode step over the 131072 cached rows of a retrieval head and over the same
    rows permuted (K and V together) agree up to the order of the fp32 sums — one bf16 ulp of the output"""
    g = torch.Generator(device=DEV).manual_seed(25)
    call, t = _segments(1, g)
    base = call()
    N = t["N"]
    perm = torch.randperm(N, generator=g, device=DEV)
    fk2, fv2 = t["fk"].clone(), t["fv"].clone()
    fk2[:N], fv2[:N] = t["fk"][:N][perm], t["fv"][:N][perm]
    got = call(fk_=fk2, fv_=fv2)
    nq = t["nf"] * t["G"]
    a, b = got[:, :nq].float(), base[:, :nq].float()
    assert ((a - b).abs() <= b.abs() * 2.0 ** -7 + 1e-3 * b.pow(2).mean().sqrt()).all(), (a - b).abs().max()
    assert torch.equal(got[:, nq:], base[:, nq:])          # the streaming heads' inputs did not change


def test_retrieval_heads_do_not_depend_on_the_chunk_size_at_128k():
    """SURVEY §8(c) property (ii): a retrieval head's row attends to keys {0 .. its position} whatever the chunking — the last
    1024 rows of a 131072-token prompt computed as the tail of a 16384-row chunk (past 114688) and as a 1024-row chunk of their
    own (past 130048) agree within the two runs' independent bf16-P rounding noise"""
    be = _be()
    g = torch.Generator(device=DEV).manual_seed(27)
    nf, G, T, big, small = 2, 4, 131072, 16384, 1024
    scale = D ** -0.5
    q = _randn((big, nf * G, D), g)
    fk, fv = _pool(nf, T, g), _pool(nf, T, g)

    def run(S):
        past = T - S
        out = torch.empty(S, nf * G, D, device=DEV, dtype=torch.bfloat16)
        be.attention(q[big - S:], out, G, (nf, 0, (fk[:past], fv[:past]), (fk[past:], fv[past:])), None, scale)
        torch.cuda.synchronize()
        return out[S - small:].float()

    a, b = run(big), run(small)
    rms = b.pow(2).mean().sqrt()
    assert (a - b).pow(2).mean().sqrt() <= 3.6e-3 * rms          # sqrt(2) x the 2.5e-3 bar each run holds against exact P
    assert ((a - b).abs() <= 2.0 ** -6 * b.abs() + 3e-2 * rms).all(), (a - b).abs().max()     # (~9 sigma of that noise)
