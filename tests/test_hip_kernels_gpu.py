"""GPU parity: every HIP kernel, called through the C ABI (ctypes), against the CPU oracle.

Tolerances (floating point path):
  * data movement (append / compress): bit-exact;
  * RoPE / RMSNorm: bf16 outputs equal to the oracle's up to 1 bf16 ulp on < 1 % of elements
    (device sincos / rsqrt may differ from the host's by an fp32 ulp before the bf16 rounding);
  * attention: helpers.attn_close — 1e-3 relative + one bf16 ulp of the value (+1e-3*rms floor).
"""
import itertools

import pytest
import torch

from helpers import ShapeModel, attn_close, heads_from_counts, with_rounded
from oracle.duo_oracle import (
    StaticCacheRef,
    flash_attn_func_ref,
    rmsnorm_ref,
    rope_ref,
    static_forward_ref,
)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _hip():
    from duo_attn import _hip

    _hip.load_library()
    return _hip


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(torch.bfloat16)


def _ulp_close(ours, ref, what, max_frac=0.01):
    """bf16 tensors equal except <= max_frac of elements, those off by at most one bf16 ulp."""
    o, r = ours.cpu().float(), ref.cpu().float()
    assert o.shape == r.shape, (o.shape, r.shape)
    if o.numel() == 0:
        return
    diff = (o - r).abs()
    tol = torch.clamp(torch.maximum(o.abs(), r.abs()) * 2.0 ** -7, min=1e-5)
    assert (diff <= tol).all(), f"{what}: max diff {diff.max():.3e} exceeds one bf16 ulp"
    frac = (diff > 0).float().mean().item()
    assert frac <= max_frac, f"{what}: {frac:.4%} elements differ"


# ----------------------------------------------------------------------------- library
def test_library_is_hip_gfx950():
    h = _hip()
    lib = h.load_library()
    assert lib.duo_target_arch() == b"gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ----------------------------------------------------------------------------- RoPE
@pytest.mark.parametrize("pos0,scale,theta", [(0, 1.0, 1e4), (1000, 8.0, 1e4), (131072, 1.0, 3580165449.0),
                                              (1_000_000, 1.0, 1e6)])
def test_rope(pos0, scale, theta):
    h = _hip()
    g = torch.Generator().manual_seed(pos0 + 1)
    S, Hq, Hkv = 37, 8, 2
    q, k = _rand((S, Hq, D), g), _rand((S, Hkv, D), g)
    # strided views like the projections' [S, H*D] outputs
    qd, kd = q.to(DEV), k.to(DEV)
    h.rope_inplace(qd, kd, pos0, scale, theta)
    _ulp_close(qd, rope_ref(q, pos0, scale, theta), "rope q")
    _ulp_close(kd, rope_ref(k, pos0, scale, theta), "rope k")


def test_rope_fp16():
    """duo_rope_inplace_f16 (the INT4-KV path's fp16 model): same angles, fp16 in/out — within one fp16 ulp of the
    oracle's fp32 rotation on a small fraction of elements (device vs host sincos last-bit differences)."""
    h = _hip()
    g = torch.Generator().manual_seed(5)
    S, Hq, Hkv = 41, 4, 2
    q = torch.randn(S, Hq, D, generator=g).to(torch.float16)
    k = torch.randn(S, Hkv, D, generator=g).to(torch.float16)
    for pos0, theta in ((0, 1e4), (70000, 500000.0)):
        qd, kd = q.to(DEV), k.to(DEV)
        h.rope_inplace(qd, kd, pos0, 1.0, theta)
        for got, x in ((qd, q), (kd, k)):
            want = rope_ref(x, pos0, 1.0, theta)
            diff = (got.cpu().float() - want.float()).abs()
            assert (diff <= want.float().abs() * 2.0 ** -10 + 1e-6).all()
            assert (diff > 0).float().mean() < 0.03


# ----------------------------------------------------------------------------- append / compress
@pytest.mark.parametrize("head_major", [True, False])
def test_kv_append(head_major):
    h = _hip()
    g = torch.Generator().manual_seed(3)
    T, nh, S, row0 = 300, 3, 70, 123
    ks, vs = _rand((S, 5, D), g), _rand((S, 5, D), g)        # source has 5 heads, we append heads 1..3
    if head_major:
        kp = torch.zeros(nh, T, D, dtype=torch.bfloat16, device=DEV).permute(1, 0, 2)
        vp = torch.zeros(nh, T, D, dtype=torch.bfloat16, device=DEV).permute(1, 0, 2)
    else:
        kp = torch.zeros(T, nh, D, dtype=torch.bfloat16, device=DEV)
        vp = torch.zeros(T, nh, D, dtype=torch.bfloat16, device=DEV)
    h.kv_append(ks.to(DEV)[:, 1:4], vs.to(DEV)[:, 1:4], kp, vp, row0)
    ek = torch.zeros(T, nh, D, dtype=torch.bfloat16)
    ev = torch.zeros(T, nh, D, dtype=torch.bfloat16)
    ek[row0:row0 + S] = ks[:, 1:4]
    ev[row0:row0 + S] = vs[:, 1:4]
    assert torch.equal(kp.cpu(), ek) and torch.equal(vp.cpu(), ev)


@pytest.mark.parametrize(
    "sink,recent,cur,n_new",
    [(128, 256, 0, 1), (128, 256, 0, 300), (128, 256, 0, 1000), (128, 256, 10, 5), (128, 256, 383, 1),
     (128, 256, 384, 1), (128, 256, 384, 300), (128, 256, 200, 4096), (4, 8, 12, 1), (4, 8, 3, 2),
     (16, 64, 80, 7), (64, 256, 100, 400), (0, 8, 8, 3), (4, 300, 304, 1), (4, 300, 304, 777)],
)
@pytest.mark.parametrize("head_major", [True, False])
def test_stream_compress(sink, recent, cur, n_new, head_major):
    from oracle.duo_oracle import OracleBackend

    h = _hip()
    g = torch.Generator().manual_seed(cur * 7 + n_new)
    W, nh = sink + recent, 3
    pool_k, pool_v = _rand((W, nh, D), g), _rand((W, nh, D), g)
    new_k, new_v = _rand((n_new, nh, D), g), _rand((n_new, nh, D), g)
    if head_major:
        kp = pool_k.permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)
        vp = pool_v.permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)
    else:
        kp, vp = pool_k.to(DEV), pool_v.to(DEV)
    n = h.stream_compress(kp, vp, new_k.to(DEV), new_v.to(DEV), cur, sink, recent)
    ek, ev = pool_k.clone(), pool_v.clone()
    en = OracleBackend().stream_compress(ek, ev, new_k, new_v, cur, sink, recent)
    assert n == en
    assert torch.equal(kp.cpu()[:n], ek[:n]) and torch.equal(vp.cpu()[:n], ev[:n])
    # rows past the live length are untouched
    assert torch.equal(kp.cpu()[n:], pool_k[n:])


# ----------------------------------------------------------------------------- RMSNorm
@pytest.mark.parametrize("rows,hidden", [(1, 4096), (33, 4096), (5, 512), (7, 1024 + 8)])
def test_rmsnorm(rows, hidden):
    h = _hip()
    g = torch.Generator().manual_seed(rows)
    x, w = _rand((rows, hidden), g, 3.0), _rand((hidden,), g)
    y = h.rmsnorm(x.to(DEV), w.to(DEV), 1e-5)
    _ulp_close(y, rmsnorm_ref(x, w, 1e-5), "rmsnorm", max_frac=0.02)


# ----------------------------------------------------------------------------- attention
def _make_pool(T, nh, gen, head_major):
    k, v = _rand((T, nh, D), gen), _rand((T, nh, D), gen)
    if head_major:
        kd = k.permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)
        vd = v.permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)
    else:
        kd, vd = k.to(DEV), v.to(DEV)
    return k, v, kd, vd


def _attention_case(S, group, nf, ns, lenA_full, lenA_stream, head_major, seed, first_chunk=False):
    """Build inputs, run the HIP path through the backend adapter, return (ours, oracle fp32).

    The reference is always the oracle's exact-P form.  The MFMA prefill kernel rounds P to bf16 before
    P.V like FA2 does and gets the matching error budget (helpers.attn_close); the scalar-FMA decode
    kernel keeps P in fp32 (never less accurate than FA2) and gets none."""
    kw = dict(round_p=False, out_dtype=torch.float32, return_budget=True)
    rkw = dict(round_p=True, out_dtype=torch.bfloat16)       # the reference's own arithmetic (FA2): helpers.with_rounded
    from duo_attn.backend import HipBackend

    g = torch.Generator().manual_seed(seed)
    Hq = (nf + ns) * group
    q = _rand((S, Hq, D), g)
    k_new, v_new = _rand((S, nf + ns, D), g), _rand((S, nf + ns, D), g)
    qd, knd, vnd = q.to(DEV), k_new.to(DEV), v_new.to(DEV)
    out = torch.full((S, Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    be = HipBackend()
    scale = D ** -0.5
    ref = torch.empty(S, Hq, D, dtype=torch.float32)
    bud = torch.empty(S, Hq, D, dtype=torch.float32)
    rnd = torch.empty(S, Hq, D, dtype=torch.float32)
    if first_chunk:
        be.attention(qd, out, group, (nf + ns, 0, None, (knd, vnd)), None, scale)
        r, b = flash_attn_func_ref(q[None], k_new[None], v_new[None], **kw)
        return out, r[0], with_rounded(b[0], flash_attn_func_ref(q[None], k_new[None], v_new[None], **rkw)[0])
    full = stream = None
    if nf:
        fk, fv, fkd, fvd = _make_pool(lenA_full, nf, g, head_major)
        full = (nf, 0, (fkd, fvd) if lenA_full else None, (knd[:, :nf], vnd[:, :nf]))
        kk = torch.cat([fk, k_new[:, :nf]], 0)
        vv = torch.cat([fv, v_new[:, :nf]], 0)
        r, b = flash_attn_func_ref(q[None, :, :nf * group], kk[None], vv[None], **kw)
        ref[:, :nf * group], bud[:, :nf * group] = r[0], b[0]
        if S > 1:
            rnd[:, :nf * group] = flash_attn_func_ref(q[None, :, :nf * group], kk[None], vv[None], **rkw)[0]
    if ns:
        sk, sv, skd, svd = _make_pool(lenA_stream, ns, g, head_major)
        stream = (ns, nf * group, (skd, svd) if lenA_stream else None, (knd[:, nf:], vnd[:, nf:]))
        kk = torch.cat([sk, k_new[:, nf:]], 0)
        vv = torch.cat([sv, v_new[:, nf:]], 0)
        r, b = flash_attn_func_ref(q[None, :, nf * group:], kk[None], vv[None], **kw)
        ref[:, nf * group:], bud[:, nf * group:] = r[0], b[0]
        if S > 1:
            rnd[:, nf * group:] = flash_attn_func_ref(q[None, :, nf * group:], kk[None], vv[None], **rkw)[0]
    be.attention(qd, out, group, full, stream, scale)
    return out, ref, (with_rounded(bud, rnd) if S > 1 else None)


DECODE_CASES = [
    # group, nf, ns, N_full, n_stream
    (4, 1, 1, 1, 1), (4, 2, 6, 17, 5), (4, 4, 4, 255, 384), (4, 8, 0, 256, 0), (4, 0, 8, 0, 383),
    (4, 3, 5, 257, 384), (4, 1, 7, 1000, 384), (4, 5, 3, 5000, 384), (4, 2, 2, 40000, 384),
    (1, 8, 24, 3000, 384), (2, 3, 1, 777, 100), (8, 1, 1, 2049, 50), (3, 2, 2, 515, 30), (6, 1, 1, 300, 7),
]


@pytest.mark.parametrize("case", DECODE_CASES)
@pytest.mark.parametrize("head_major", [True, False])
def test_decode(case, head_major):
    group, nf, ns, n_full, n_stream = case
    out, ref, bud = _attention_case(1, group, nf, ns, n_full, n_stream, head_major, seed=hash(case) % 1000)
    attn_close(out, ref, f"decode {case} hm={head_major}", bud)


FUSED_CASES = [
    # group, nf, ns, full_len, str_len, sink, recent, pos
    (4, 4, 4, 1000, 384, 128, 256, 1000), (4, 1, 7, 5000, 384, 128, 256, 131071), (4, 8, 0, 300, 12, 4, 8, 300),
    (4, 0, 8, 0, 383, 128, 256, 77), (8, 1, 1, 2049, 11, 4, 8, 2049), (1, 3, 5, 64, 5, 4, 8, 1_000_000),
    (2, 2, 2, 0, 0, 4, 8, 0), (4, 2, 6, 40000, 384, 128, 256, 40000), (3, 2, 1, 515, 12, 4, 8, 515),
]


@pytest.mark.parametrize("two_launch", [False, True])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_decode_layer_fused(case, two_launch):
    """duo_decode_step_bf16 (ONE launch: scan with RoPE + retrieval append folded in, merge + streaming update
    behind per-head arrival tickets) and duo_decode_layer_bf16 (the same as two launches) against the oracle's
    step-by-step restatement of llama.py:332-425 for q_len == 1."""
    from oracle.duo_oracle import OracleBackend

    group, nf, ns, full_len, str_len, sink, recent, pos = case
    h = _hip()
    g = torch.Generator().manual_seed(hash(case) % 1000)
    nkv, Hq, W = nf + ns, (nf + ns) * group, sink + recent
    theta, rscale = 3580165449.0, 1.0
    q, k, v = _rand((Hq, D), g), _rand((nkv, D), g), _rand((nkv, D), g)
    cap = full_len + 3
    fk, fv, fkd, fvd = _make_pool(cap, max(nf, 1), g, True)
    sk, sv, skd, svd = _make_pool(W, max(ns, 1), g, True)
    fk, fv, fkd, fvd = fk[:, :nf], fv[:, :nf], fkd[:, :nf], fvd[:, :nf]
    sk, sv, skd, svd = sk[:, :ns], sv[:, :ns], skd[:, :ns], svd[:, :ns]
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out = torch.full((Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    n = h.decode_layer(qd, kd, vd, out, nf, fkd, fvd, full_len, skd, svd, str_len, sink, recent, pos, rscale,
                       theta, D ** -0.5, two_launch=two_launch)
    assert int(h.decode_tickets(torch.device(DEV)).abs().sum()) == 0     # re-armed (and no give-up flag)
    # inputs are read-only
    assert torch.equal(qd.cpu(), q) and torch.equal(kd.cpu(), k) and torch.equal(vd.cpu(), v)

    qr = rope_ref(q[None], pos, rscale, theta)[0]
    kr = rope_ref(k[None], pos, rscale, theta)[0]
    kw = dict(round_p=False, out_dtype=torch.float32)
    ref = torch.empty(Hq, D, dtype=torch.float32)
    if nf:
        kk = torch.cat([fk[:full_len], kr[None, :nf]], 0)
        vv = torch.cat([fv[:full_len], v[None, :nf]], 0)
        ref[:nf * group] = flash_attn_func_ref(qr[None, None, :nf * group], kk[None], vv[None], **kw)[0, 0]
        # the pool gained the rotated key / the value at row full_len; everything else untouched
        _ulp_close(fkd.cpu()[full_len], kr[:nf], "appended k")
        assert torch.equal(fvd.cpu()[full_len], v[:nf])
        assert torch.equal(fkd.cpu()[:full_len], fk[:full_len]) and torch.equal(fkd.cpu()[full_len + 1:], fk[full_len + 1:])
        assert torch.equal(fvd.cpu()[:full_len], fv[:full_len]) and torch.equal(fvd.cpu()[full_len + 1:], fv[full_len + 1:])
    T = str_len + 1
    assert n == min(T, W)
    if ns:
        kk = torch.cat([sk[:str_len], kr[None, nf:]], 0)
        vv = torch.cat([sv[:str_len], v[None, nf:]], 0)
        ref[nf * group:] = flash_attn_func_ref(qr[None, None, nf * group:], kk[None], vv[None], **kw)[0, 0]
        ek, ev = sk.clone(), sv.clone()
        en = OracleBackend().stream_compress(ek, ev, kr[None, nf:], v[None, nf:], str_len, sink, recent)
        assert en == n
        # V rows and the old K rows move bit-exactly; the one new K row is rotated on the device
        assert torch.equal(svd.cpu()[:n], ev[:n])
        assert torch.equal(skd.cpu()[:n - 1], ek[:n - 1])
        _ulp_close(skd.cpu()[n - 1], ek[n - 1], "new streaming k")
    # the device rotates q/k with its own sincos: allow the 1-ulp bf16 input differences through the
    # attention tolerance (rms floor)
    attn_close(out, ref, f"fused decode {case}", None)


@pytest.mark.parametrize("group,nf,ns,full_len", [(4, 4, 4, 70000), (4, 1, 7, 131072), (4, 0, 8, 0), (1, 8, 24, 9000),
                                                  (2, 3, 5, 33333), (4, 8, 0, 20000)])
def test_single_launch_step_is_bit_identical_to_two_launches(group, nf, ns, full_len):
    """Same partials, same merge arithmetic: the one-launch step must reproduce the two-launch step bit for
    bit — outputs and both pools — also when launched back to back many times (ticket re-arming, the merge
    reading partials other CUs wrote into the SAME workspace addresses one launch earlier)."""
    h = _hip()
    g = torch.Generator().manual_seed(full_len % 977 + nf)
    sink, recent = 128, 256
    nkv, Hq, W = nf + ns, (nf + ns) * group, sink + recent
    cap = full_len + 40

    def fresh():
        gg = torch.Generator().manual_seed(5)
        _, _, fkd, fvd = _make_pool(cap, max(nf, 1), gg, True)
        _, _, skd, svd = _make_pool(W, max(ns, 1), gg, True)
        return fkd[:, :nf], fvd[:, :nf], skd[:, :ns], svd[:, :ns]

    steps = 24
    qs = [_rand((Hq, D), g).to(DEV) for _ in range(steps)]
    ks = [_rand((nkv, D), g).to(DEV) for _ in range(steps)]
    vs = [_rand((nkv, D), g).to(DEV) for _ in range(steps)]
    results = []
    for two in (True, False):
        fkd, fvd, skd, svd = fresh()
        outs = []
        str_len = W - 5                       # crosses the fill -> slide transition of the streaming pool
        for i in range(steps):                # back to back, no host sync in between
            out = torch.empty(Hq, D, dtype=torch.bfloat16, device=DEV)
            str_len = h.decode_layer(qs[i], ks[i], vs[i], out, nf, fkd, fvd, full_len + i, skd, svd, str_len, sink,
                                     recent, full_len + i, 1.0, 5e5, D ** -0.5, two_launch=two)
            outs.append(out)
        torch.cuda.synchronize()
        results.append((torch.stack(outs).cpu(), fkd.cpu().clone(), fvd.cpu().clone(), skd.cpu().clone(), svd.cpu().clone()))
    assert int(h.decode_tickets(torch.device(DEV)).abs().sum()) == 0
    for a, b in zip(*results):
        assert torch.equal(a, b)


PREFILL_CASES = [
    # S, group, nf, ns, lenA_full, lenA_stream
    (2, 4, 1, 1, 3, 2), (31, 4, 1, 1, 64, 100), (64, 4, 2, 2, 5, 384), (100, 4, 1, 3, 1000, 384),
    (256, 4, 1, 1, 256, 384), (257, 4, 2, 0, 300, 0), (300, 4, 0, 2, 0, 384), (513, 1, 2, 2, 77, 10),
    (1000, 4, 1, 1, 1000, 384), (129, 2, 1, 1, 63, 65), (700, 8, 1, 0, 129, 0),
]


# which prefill kernel a launch lands on: 0 = automatic (4-wave x 64-row kernel unless the launcher splits the key
# range), 128 = the 8-wave x 32-row kernel everywhere, 256 = no key-range split -> always the 4-wave kernel
PREFILL_KERNELS = [0, 128, 256]


@pytest.fixture(params=PREFILL_KERNELS, ids=["auto", "w8x32", "w4x64"])
def prefill_kernel(request):
    h = _hip()
    h.set_debug_flags(request.param)
    yield request.param
    h.set_debug_flags(0)


@pytest.mark.parametrize("case", PREFILL_CASES)
@pytest.mark.parametrize("head_major", [True, False])
def test_prefill_later_chunk(case, head_major, prefill_kernel):
    S, group, nf, ns, la, ls = case
    out, ref, bud = _attention_case(S, group, nf, ns, la, ls, head_major, seed=hash(case) % 1000)
    attn_close(out, ref, f"prefill {case} hm={head_major}", bud)


@pytest.mark.parametrize("S,group,nkv", [(2, 4, 2), (65, 4, 2), (256, 4, 1), (300, 1, 4), (1025, 4, 2)])
def test_prefill_first_chunk(S, group, nkv, prefill_kernel):
    out, ref, bud = _attention_case(S, group, nkv, 0, 0, 0, True, seed=S, first_chunk=True)
    attn_close(out, ref, f"first chunk S={S}", bud)


@pytest.mark.parametrize("S,lenA,lenB,group", [(300, 100, 500, 4), (256, 0, 1024, 4), (77, 384, 78, 1), (512, 1000, 513, 2),
                                               (512, 1000, 2048, 2), (640, 0, 1700, 1)])
def test_prefill_query_block_shorter_than_segment_b(S, lenA, lenB, group, prefill_kernel):
    """segment B longer than the query block: the S queries are its last S rows (bottom-right alignment,
    flash_attn_func with seqlen_q < seqlen_k) — a chunk processed in row blocks."""
    from duo_attn.backend import HipBackend

    g = torch.Generator().manual_seed(S + lenB)
    q = _rand((S, group, D), g)
    ka, va = _rand((lenA, 1, D), g), _rand((lenA, 1, D), g)
    kb, vb = _rand((lenB, 1, D), g), _rand((lenB, 1, D), g)
    out = torch.full((S, group, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    segA = (ka.to(DEV), va.to(DEV)) if lenA else None
    HipBackend().attention(q.to(DEV), out, group, (1, 0, segA, (kb.to(DEV), vb.to(DEV))), None, D ** -0.5)
    ref, bud = flash_attn_func_ref(q[None], torch.cat([ka, kb])[None], torch.cat([va, vb])[None], round_p=False,
                                   out_dtype=torch.float32, return_budget=True)
    attn_close(out, ref[0], f"row block S={S} lenB={lenB}", bud[0])


def test_prefill_key_range_splits_agree_with_single_pass(prefill_kernel):
    """duo_attn_prefill_ws_bf16 (workspace: the retrieval class may be split over key ranges, partials merged
    by a second launch) against duo_attn_prefill_bf16 (one workgroup walks all of a q tile's keys): same
    attention, fp32 summation order aside.  Shape chosen so that the launcher does split (8 long workgroups)."""
    from ctypes import byref

    h = _hip()
    lib = h.load_library()
    g = torch.Generator().manual_seed(77)
    S, group, lenA = 300, 4, 5000
    q = _rand((S, group, D), g).to(DEV)
    ka, va = _rand((lenA, 1, D), g).to(DEV), _rand((lenA, 1, D), g).to(DEV)
    kb, vb = _rand((S, 1, D), g).to(DEV), _rand((S, 1, D), g).to(DEV)
    cls = h.make_class(1, 0, h.make_seg(ka, va), h.make_seg(kb, vb))
    out_split = torch.empty_like(q)
    h.attn_prefill(q, out_split, group, cls, None, D ** -0.5)
    out_single = torch.empty_like(q)
    rc = lib.duo_attn_prefill_bf16(q.data_ptr(), q.stride(0), q.stride(1), out_single.data_ptr(), out_single.stride(0),
                                   out_single.stride(1), S, group, byref(cls), None, D ** -0.5, D,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    a, b = out_split.float().cpu(), out_single.float().cpu()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    # The two differ by more than summation order: each pass rounds P to bf16 relative to ITS running
    # maximum, so the P-rounding noise (budgeted in attn_close) is drawn twice.  Both must meet the oracle bar,
    # and each other within twice that noise.
    ref, bud = flash_attn_func_ref(q.cpu()[None], torch.cat([ka, kb]).cpu()[None], torch.cat([va, vb]).cpu()[None],
                                   round_p=False, out_dtype=torch.float32, return_budget=True)
    attn_close(out_split, ref[0], "split", bud[0])
    attn_close(out_single, ref[0], "single", bud[0])
    rms = b.pow(2).mean().sqrt()
    assert (a - b).pow(2).mean().sqrt() <= 5e-3 * rms


@pytest.mark.parametrize("eight_wave", [False, True], ids=["w4x64", "w8x32"])
@pytest.mark.parametrize("ksplit", [2, 3, 5, 8])
@pytest.mark.parametrize("S,lenA", [(700, 3000), (256, 0), (1024, 200)])
def test_prefill_forced_key_range_splits(S, lenA, ksplit, eight_wave):
    """Every workgroup of the retrieval class walks 1/k of its tile sequence (debug bits 12-15 force k): ranges that
    start inside the cached segment, on its partial last tile, inside the chunk's own tiles (rows that see no key of
    their range leave m = -inf, l = 0), bulk runs that begin at a split boundary — merged partials against the oracle."""
    h = _hip()
    g = torch.Generator().manual_seed(S + lenA + ksplit)
    group = 2
    q = _rand((S, group, D), g)
    ka, va = _rand((lenA, 1, D), g), _rand((lenA, 1, D), g)
    kb, vb = _rand((S, 1, D), g), _rand((S, 1, D), g)
    out = torch.full((S, group, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    qd, kad, vad, kbd, vbd = (t.to(DEV) for t in (q, ka, va, kb, vb))     # the segments hold raw pointers: keep these alive
    cls = h.make_class(1, 0, h.make_seg(kad, vad), h.make_seg(kbd, vbd))
    h.set_debug_flags((ksplit << 12) | (128 if eight_wave else 0))
    try:
        h.attn_prefill(qd, out, group, cls, None, D ** -0.5)
        torch.cuda.synchronize()
    finally:
        h.set_debug_flags(0)
    ref, bud = flash_attn_func_ref(q[None], torch.cat([ka, kb])[None], torch.cat([va, vb])[None], round_p=False,
                                   out_dtype=torch.float32, return_budget=True)
    # each range rounds P relative to ITS running maximum: the P-rounding budget is drawn once per range that matters
    attn_close(out, ref[0], f"forced {ksplit} splits S={S} lenA={lenA}", bud[0])


@pytest.mark.parametrize("flags", [0, 128, 1024], ids=["w4x64", "w8x32", "plain-order"])
@pytest.mark.parametrize("k0,k1", [(2, 2), (5, 3), (1, 4), (15, 1), (3, 2)])
@pytest.mark.parametrize("S,group,nf,ns,lenA", [(700, 4, 3, 5, 3000), (520, 2, 1, 2, 1500), (300, 3, 2, 0, 2000)])
def test_prefill_key_range_splits_of_both_head_classes(S, group, nf, ns, lenA, k0, k1, flags):
    """Round 6: the launcher may cut BOTH classes over key ranges (debug bits 12-15 / 16-19 force the counts), up to 16 pieces,
    in the XCD-aware block order (virtual heads = kv head x piece) or the plain one (bit 10), on either kernel: ragged head
    counts, a streaming window next to a long pool, pieces that begin inside the window, on its partial last tile or inside
    the chunk's own tiles — merged partials against the oracle."""
    h = _hip()
    h.set_debug_flags((k0 << 12) | (k1 << 16) | flags)
    try:
        out, ref, bud = _attention_case(S, group, nf, ns, lenA, 330, True, seed=S + 7 * k0 + k1)
        torch.cuda.synchronize()
        plan = h.last_prefill_plan()
    finally:
        h.set_debug_flags(0)
    assert plan[0] == (k0 if nf else 1) and plan[1] == (k1 if ns else 1), plan
    attn_close(out, ref, f"pieces ({k0}, {k1}) S={S} nf={nf} ns={ns} flags={flags}", bud)


@pytest.mark.parametrize("flags", [0, 1 << 21], ids=["chained", "separate-runs"])
@pytest.mark.parametrize("S,lenA,lenS,k0", [(1024, 1024, 384, 0), (700, 2048, 384, 0), (520, 192, 128, 0), (1500, 128, 64, 0), (900, 3072, 384, 3),
                                            (2048, 0, 0, 0), (640, 64, 384, 0)])
def test_prefill_bulk_run_across_the_pool_chunk_boundary(S, lenA, lenS, k0, flags):
    """Round 6: when the cached segment ends on a 64-key tile boundary the kernel's first bulk run CHAINS into the chunk's own
    tiles (the LDS-DMA source of tile t+2 switches from pool to chunk two tiles before the boundary) instead of dropping to the
    general tile form around it; debug bit 21 keeps the two runs separate.  Pools of a whole number of tiles from one tile up, a
    run that STARTS behind the switch point (no pool: the first chunk's shape; a one-tile pool), key-range pieces that begin
    inside either segment — both forms against the oracle, and against each other within the two forms' rounding."""
    h = _hip()
    outs = []
    h.set_debug_flags(flags | (k0 << 12))
    try:
        out, ref, bud = _attention_case(S, 4, 2 if lenA else 0, 2, lenA, lenS, True, seed=S + lenA, first_chunk=(lenA == 0 and lenS == 0))
        torch.cuda.synchronize()
    finally:
        h.set_debug_flags(0)
    attn_close(out, ref, f"bulk run across the boundary S={S} lenA={lenA} lenS={lenS} k0={k0} flags={flags}", bud)


def test_split_launches_back_to_back_through_one_workspace():
    """the same workspace serves launch after launch: five split launches with DIFFERENT inputs back to back (no
    synchronisation in between), then the first one again — every output against the oracle, and the repeated launch bit-equal
    to its first run (a merge pass that read the partials of the launch before would show here)"""
    from duo_attn.backend import HipBackend

    h = _hip()
    be = HipBackend()
    S, group, nf, lenA = 520, 2, 2, 2500
    cases = []
    for i in range(5):
        g = torch.Generator().manual_seed(100 + i)
        q = _rand((S, nf * group, D), g)
        ka, va, kb, vb = _rand((lenA, nf, D), g), _rand((lenA, nf, D), g), _rand((S, nf, D), g), _rand((S, nf, D), g)
        cases.append((q, ka, va, kb, vb, [t.to(DEV) for t in (q, ka, va, kb, vb)]))
    outs = [torch.full((S, nf * group, D), float("nan"), dtype=torch.bfloat16, device=DEV) for _ in range(6)]
    h.set_debug_flags((4 << 12))
    try:
        for i, (_, _, _, _, _, (qd, kad, vad, kbd, vbd)) in enumerate(cases + cases[:1]):
            be.attention(qd, outs[i], group, (nf, 0, (kad, vad), (kbd, vbd)), None, D ** -0.5)
        torch.cuda.synchronize()
        assert h.last_prefill_plan()[0] == 4
    finally:
        h.set_debug_flags(0)
    assert torch.equal(outs[5], outs[0])
    for i, (q, ka, va, kb, vb, _) in enumerate(cases):
        ref, bud = flash_attn_func_ref(q[None], torch.cat([ka, kb])[None], torch.cat([va, vb])[None], round_p=False,
                                       out_dtype=torch.float32, return_budget=True)
        attn_close(outs[i], ref[0], f"split launches back to back, launch {i}", bud[0])


def test_prefill_row_block_launch_is_planned_and_matches():
    """the layer pipeline's launch shape (queries = the LAST rows of segment B, duo_static_attention_row_block): the planner
    splits both classes by itself (no forced count) and the result equals the oracle's attention over pool ++ chunk rows"""
    from duo_attn.backend import HipBackend

    h = _hip()
    g = torch.Generator().manual_seed(11)
    S, r1, group, nf, ns, lenA = 512, 1536, 4, 1, 1, 9000
    q = _rand((S, (nf + ns) * group, D), g)
    kb, vb = _rand((r1, nf + ns, D), g), _rand((r1, nf + ns, D), g)
    ka, va = _rand((lenA, nf, D), g), _rand((lenA, nf, D), g)
    sk, sv = _rand((300, ns, D), g), _rand((300, ns, D), g)
    dev = [t.to(DEV) for t in (q, ka, va, kb, vb, sk, sv)]
    qd, kad, vad, kbd, vbd, skd, svd = dev
    out = torch.full(q.shape, float("nan"), dtype=torch.bfloat16, device=DEV)
    HipBackend().attention(qd, out, group, (nf, 0, (kad, vad), (kbd[:, :nf], vbd[:, :nf])),
                           (ns, nf * group, (skd, svd), (kbd[:, nf:], vbd[:, nf:])), D ** -0.5)
    torch.cuda.synchronize()
    plan = h.last_prefill_plan()
    assert plan[0] > 1 and plan[2] < plan[3], plan        # 8 long workgroups on 256 CUs: split, and estimated cheaper
    kw = dict(round_p=False, out_dtype=torch.float32, return_budget=True)
    for lo, hi, kk, vv in ((0, nf * group, torch.cat([ka, kb[:, :nf]]), torch.cat([va, vb[:, :nf]])),
                           (nf * group, (nf + ns) * group, torch.cat([sk, kb[:, nf:]]), torch.cat([sv, vb[:, nf:]]))):
        ref, bud = flash_attn_func_ref(q[None, :, lo:hi], kk[None], vv[None], **kw)
        attn_close(out[:, lo:hi], ref[0], f"row block, heads {lo}:{hi}", bud[0])


def test_prefill_without_transpose_read_matches():
    """8-wave kernel: ds_read_b64_tr_b16 path == scalar LDS gather path (debug flag bit 0), bit for bit."""
    h = _hip()
    case = (300, 4, 1, 1, 200, 384)
    h.set_debug_flags(128)            # bit 7: the 8-wave kernel (the gather path only exists there)
    try:
        out_tr, ref, bud = _attention_case(*case, True, seed=5)
        h.set_debug_flags(128 | 1)
        out_gather, _, _ = _attention_case(*case, True, seed=5)
    finally:
        h.set_debug_flags(0)
    assert torch.equal(out_tr.cpu(), out_gather.cpu())
    attn_close(out_gather, ref, "prefill (gather V path)", bud)


def test_softmax_rescale_branch_spike(prefill_kernel):
    """A key late in the sequence that dominates every earlier score forces the online-softmax
    rescale path (m jumps by >> 8 in the last tiles)."""
    from duo_attn.backend import HipBackend

    g = torch.Generator().manual_seed(11)
    S, group = 320, 4
    q = _rand((S, group, D), g)
    k, v = _rand((S, 1, D), g), _rand((S, 1, D), g)
    k[300, 0] = (q[310, 1].float() * 4).to(torch.bfloat16)    # spike for (row 310, head 1) at key 300
    out = torch.empty(S, group, D, dtype=torch.bfloat16, device=DEV)
    HipBackend().attention(q.to(DEV), out, group, (1, 0, None, (k.to(DEV), v.to(DEV))), None, D ** -0.5)
    ref, bud = flash_attn_func_ref(q[None], k[None], v[None], round_p=False, out_dtype=torch.float32,
                                   return_budget=True)
    attn_close(out, ref[0], "spike", bud[0])


def test_rescale_inside_cached_segment_both_row_blocks(prefill_kernel):
    """Keys deep inside the cached segment (the 4-wave kernel's bulk tiles: full 64-key tiles, no masks) that
    dominate every earlier score of chosen query rows — rows of the first AND the second 32-row block of a wave,
    and a second, even larger spike later — force the deferred-rescale branch (O, the row sum and the running
    maximum rescaled; in the 4-wave kernel O lives in asm-owned accumulator registers) at a chosen tile for each."""
    from duo_attn.backend import HipBackend

    g = torch.Generator().manual_seed(13)
    S, group, lenA = 200, 4, 1100
    q = _rand((S, group, D), g)
    kp, vp = _rand((lenA, 1, D), g), _rand((lenA, 1, D), g)
    kn, vn = _rand((S, 1, D), g), _rand((S, 1, D), g)
    for row, head, key, gain in ((5, 0, 300, 3.0), (40, 1, 333, 3.0), (70, 2, 520, 3.0), (120, 3, 700, 3.0),
                                 (5, 0, 900, 8.0), (40, 1, 64 * 3 + 1, 2.5)):
        kp[key, 0] = (q[row, head].float() * gain).to(torch.bfloat16)
    out = torch.empty(S, group, D, dtype=torch.bfloat16, device=DEV)
    HipBackend().attention(q.to(DEV), out, group, (1, 0, (kp.to(DEV), vp.to(DEV)), (kn.to(DEV), vn.to(DEV))), None, D ** -0.5)
    kk, vv = torch.cat([kp, kn], 0), torch.cat([vp, vn], 0)
    ref, bud = flash_attn_func_ref(q[None], kk[None], vv[None], round_p=False, out_dtype=torch.float32, return_budget=True)
    attn_close(out, ref[0], "rescale in cached tiles", bud[0])


@pytest.mark.parametrize("lenA", [0, 640, 1100])
def test_rescale_at_every_position_of_a_bulk_run(lenA, prefill_kernel):
    """The 4-wave kernel's bulk tiles run a skewed schedule (block B of tile t-1 finishes inside tile t; first tile
    of a run, steady tiles, drain — duo_prefill_w64_bulk.inc), for runs in the cached segment AND among the chunk's own
    fully visible tiles.  One moderate spike per chosen query row — strong enough to pass the deferred-rescale
    threshold (score ~9.5 against a running max of ~2.5 + 5.5), weak enough that the other keys keep ~15 % of the row's
    weight — with the spike keys walking over every tile the row sees in full: first / middle / last tile of each run,
    the general tiles between runs, rows of both 32-row blocks of every wave."""
    from duo_attn.backend import HipBackend

    g = torch.Generator().manual_seed(100 + lenA)
    S, group = 1024, 2
    q = _rand((S, group, D), g)
    kp, vp = _rand((lenA, 1, D), g), _rand((lenA, 1, D), g)
    kn, vn = _rand((S, 1, D), g), _rand((S, 1, D), g)
    nA = (lenA + 63) // 64
    used = set()
    n_spikes = 0
    for i, row in enumerate(range(256, S, 3)):          # q tiles 1..3, every third row: both blocks of all four waves
        head = i % group
        tiles_seen = nA + row // 64                       # tiles this row sees in full (cached + chunk-own)
        tile = (i * 5) % tiles_seen
        key = tile * 64 + (row * 7) % 64
        if key in used or (tile == nA - 1 and lenA % 64 and key >= lenA):
            continue
        used.add(key)
        gain = 0.85 if i % 2 else 1.2
        vec = (q[row, head].float() * gain).to(torch.bfloat16)
        if key < lenA:
            kp[key, 0] = vec
        else:
            kk_ = key - nA * 64 if lenA % 64 == 0 else None
            if kk_ is None or kk_ < 0 or kk_ > row:
                continue
            kn[kk_, 0] = vec
        n_spikes += 1
    assert n_spikes > 100
    out = torch.empty(S, group, D, dtype=torch.bfloat16, device=DEV)
    segA = (kp.to(DEV), vp.to(DEV)) if lenA else None
    HipBackend().attention(q.to(DEV), out, group, (1, 0, segA, (kn.to(DEV), vn.to(DEV))), None, D ** -0.5)
    kk, vv = torch.cat([kp, kn], 0), torch.cat([vp, vn], 0)
    ref, bud = flash_attn_func_ref(q[None], kk[None], vv[None], round_p=False, out_dtype=torch.float32, return_budget=True)
    attn_close(out, ref[0], f"rescale along bulk runs lenA={lenA}", bud[0])


# ----------------------------------------------------------------------------- whole hot path
@pytest.mark.parametrize(
    "counts,Hq,Hkv,chunks,sink,recent",
    [
        ([1, 2, 0, 4], 16, 4, (300, 129, 64), 16, 48),
        ([2, 6], 32, 8, (520, 520), 128, 256),
        ([3], 4, 4, (100, 300, 77), 4, 8),
    ],
)
def test_static_hot_path_chunked_prefill_then_decode(counts, Hq, Hkv, chunks, sink, recent):
    """duo_static_attention_core on the GPU (RoPE -> append -> split-head attention -> streaming update)
    vs the oracle's restatement of reference llama.py:309-434, chunk after chunk, then 4 decode steps
    with the benchmark's evict_last(1) protocol (reference benchmark_static.py:96-105)."""
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    L = len(counts)
    heads = heads_from_counts(counts, Hkv)
    total = sum(chunks) + 8
    cache = DuoAttentionStaticKVCache(ShapeModel(L, Hq, Hkv, D, device=DEV), heads, 1, total, sink, recent)
    ref = StaticCacheRef(L, Hkv, D, heads, 1, total, sink, recent)
    g = torch.Generator().manual_seed(42)
    theta, rscale = 3580165449.0, 1.0
    pos = 0
    steps = [(S, False) for S in chunks] + [(1, True)] * 4
    for S, evict in steps:
        for l in range(L):
            q, k, v = _rand((1, S, Hq, D), g), _rand((1, S, Hkv, D), g), _rand((1, S, Hkv, D), g)
            out = duo_static_attention_core(q.to(DEV), k.to(DEV), v.to(DEV), cache, l, pos, rscale, theta)
            exp, bud = static_forward_ref(q, k, v, ref, l, pos, rscale, theta, round_p=False,
                                          out_dtype=torch.float32, return_budget=True)
            attn_close(out, exp, f"S={S} layer={l} pos={pos}", bud if S > 1 else None)
            n, m = ref.kv_seq_len_list[l], ref.streaming_kv_seq_len_list[l]
            assert cache.kv_seq_len_list[l] == n and cache.streaming_kv_seq_len_list[l] == m
            _ulp_close(cache.full_key_states_list[l][:, :n], ref.full_key_states_list[l][:, :n], "full K pool")
            assert torch.equal(cache.full_value_states_list[l][:, :n].cpu(), ref.full_value_states_list[l][:, :n])
            _ulp_close(cache.streaming_key_states_list[l][:, :m], ref.streaming_key_states_list[l][:, :m],
                       "stream K pool")
            assert torch.equal(cache.streaming_value_states_list[l][:, :m].cpu(),
                               ref.streaming_value_states_list[l][:, :m])
            # keep both sides on identical pool contents so later steps compare attention only
            ref.full_key_states_list[l][:, :n].copy_(cache.full_key_states_list[l][:, :n].cpu())
            ref.streaming_key_states_list[l][:, :m].copy_(cache.streaming_key_states_list[l][:, :m].cpu())
        if evict:
            cache.evict_last(1)
            ref.evict_last(1)
        else:
            pos += S


def test_cpu_tensor_is_refused():
    h = _hip()
    q = torch.zeros(4, 4, D, dtype=torch.bfloat16)
    with pytest.raises(h.DuoHipError, match="no CPU fallback"):
        h.rope_inplace(q, q, 0, 1.0, 1e4)


# ----------------------------------------------------------------------------- device-side step state / graph
def _decode_loop_setup(counts, Hq, Hkv, sink, recent, prefill, max_size, seed):
    from duo_attn.patch._duo import duo_static_attention_core
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    g = torch.Generator().manual_seed(seed)
    L = len(counts)
    model = ShapeModel(L, Hq, Hkv, D, device=DEV)
    cache = DuoAttentionStaticKVCache(model, heads_from_counts(counts, Hkv), 1, max_size, sink, recent)
    mk = lambda S, h: _rand((1, S, h, D), g).to(DEV)
    for li in range(L):
        duo_static_attention_core(mk(prefill, Hq), mk(prefill, Hkv), mk(prefill, Hkv), cache, li, 0, 1.0, 1e4)
    # static step inputs (a graph replays the same buffers): per layer q, k, v of the "next token"
    qs, ks, vs = [mk(1, Hq) for _ in range(L)], [mk(1, Hkv) for _ in range(L)], [mk(1, Hkv) for _ in range(L)]
    outs = [torch.zeros(1, 1, Hq, D, dtype=torch.bfloat16, device=DEV) for _ in range(L)]

    def step():
        pos = cache.kv_seq_len
        for li in range(L):
            outs[li].copy_(duo_static_attention_core(qs[li], ks[li], vs[li], cache, li, pos, 1.0, 1e4))
        return outs

    return cache, step, outs


def _pool_snapshot(cache):
    return [t.clone() for lst in (cache.full_key_states_list, cache.full_value_states_list,
                                   cache.streaming_key_states_list, cache.streaming_value_states_list) for t in lst]


@pytest.mark.parametrize("evict", [0, 1])
def test_graph_replayed_decode_equals_eager(evict):
    """DecodeStepGraph (device-side lengths, one captured step replayed) == the eager fused decode, bit
    for bit: outputs of every layer at every step, pools and counters afterwards.  evict=1 is the
    reference's benchmark protocol, evict=0 real generation (cache grows across replays, also through
    the streaming pool's fill -> slide transition)."""
    from duo_attn.graph import DecodeStepGraph

    counts, Hq, Hkv, sink, recent, prefill, steps = [1, 3, 0, 4], 16, 4, 4, 12, 13, 9
    eager_out = []
    cache_e, step_e, outs_e = _decode_loop_setup(counts, Hq, Hkv, sink, recent, prefill, 64, seed=21)
    for _ in range(steps):
        step_e()
        if evict:
            cache_e.evict_last(evict)
        eager_out.append([o.clone() for o in outs_e])
    cache_g, step_g, outs_g = _decode_loop_setup(counts, Hq, Hkv, sink, recent, prefill, 64, seed=21)
    graph = DecodeStepGraph(cache_g, step_g, evict_after=evict)
    for s in range(steps):
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(outs_g, eager_out[s]):
            assert torch.equal(a, b), f"step {s}"
    assert cache_g.kv_seq_len_list == cache_e.kv_seq_len_list
    assert cache_g.streaming_kv_seq_len_list == cache_e.streaming_kv_seq_len_list
    dev = cache_g.device_state.cpu()
    assert dev[:, 0].tolist() == cache_e.kv_seq_len_list and dev[:, 1].tolist() == cache_e.streaming_kv_seq_len_list
    for a, b in zip(_pool_snapshot(cache_g), _pool_snapshot(cache_e)):
        assert torch.equal(a, b)


def test_graph_reused_after_clear_and_new_prefill_equals_eager():
    """the device-side lengths must follow host-side changes made outside the graph.  One captured
    graph serves a second prompt of a different length (clear, eager prefill, replay) and a rewind
    (evict_last), each time bit-equal to the eager decode of the same state."""
    from duo_attn.graph import DecodeStepGraph
    from duo_attn.patch._duo import duo_static_attention_core

    counts, Hq, Hkv, sink, recent = [1, 3, 0, 4], 16, 4, 4, 12
    cache_e, step_e, outs_e = _decode_loop_setup(counts, Hq, Hkv, sink, recent, 13, 96, seed=23)
    cache_g, step_g, outs_g = _decode_loop_setup(counts, Hq, Hkv, sink, recent, 13, 96, seed=23)
    graph = DecodeStepGraph(cache_g, step_g, evict_after=0)

    def both(n):
        for s in range(n):
            step_e()
            graph.replay()
            torch.cuda.synchronize()
            for a, b in zip(outs_g, outs_e):
                assert torch.equal(a, b), f"step {s}"
        assert cache_g.kv_seq_len_list == cache_e.kv_seq_len_list
        assert cache_g.device_state.cpu()[:, 0].tolist() == cache_e.kv_seq_len_list
        assert cache_g.device_state.cpu()[:, 1].tolist() == cache_e.streaming_kv_seq_len_list

    both(3)
    # second prompt: longer than the first, prefilled eagerly in two chunks on both caches
    g = torch.Generator().manual_seed(99)
    mk = lambda S, h: _rand((1, S, h, D), g)
    for c in (cache_e, cache_g):
        c.clear()
    for S in (29, 12):
        pos = cache_e.kv_seq_len
        for li in range(len(counts)):
            q, k, v = mk(S, Hq), mk(S, Hkv), mk(S, Hkv)
            for c in (cache_e, cache_g):
                duo_static_attention_core(q.clone().to(DEV), k.clone().to(DEV), v.clone().to(DEV), c, li, pos, 1.0, 1e4)
    both(4)
    for c in (cache_e, cache_g):     # rewind three tokens on the host side only
        c.evict_last(3)
    both(2)
    for a, b in zip(_pool_snapshot(cache_g), _pool_snapshot(cache_e)):
        assert torch.equal(a, b)


def test_device_state_add_clamps():
    h = _hip()
    st = torch.tensor([[5, 3, 5, 0], [0, 12, 7, 0], [100, 11, 100, 0]], dtype=torch.int32, device=DEV)
    h.decode_state_add(st, 1, 1, 1, 12)
    assert st.cpu().tolist() == [[6, 4, 6, 0], [1, 12, 8, 0], [101, 12, 101, 0]]
    h.decode_state_add(st, -7, -7, -7, 12)
    assert st.cpu().tolist() == [[0, 0, 0, 0], [0, 5, 1, 0], [94, 5, 94, 0]]


def test_row_blocks_equal_whole_chunk_on_gpu():
    """duo_static_attention_row_block with the HIP kernels: chunks processed in row blocks (512 rows; the
    prefill kernel then runs with segment B longer than its query block, and with key-range splits) against
    the same chunks processed whole — outputs within the P-rounding noise, pools and counters identical."""
    from duo_attn.patch._duo import duo_static_attention_core, duo_static_attention_row_block
    from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

    counts, Hq, Hkv, sink, recent = [1, 3, 0, 4], 16, 4, 16, 48
    chunks, rows = [1536, 1024, 700], 512
    g = torch.Generator().manual_seed(31)
    mk = lambda S, h: _rand((1, S, h, D), g)
    data = [[(mk(S, Hq), mk(S, Hkv), mk(S, Hkv)) for _ in counts] for S in chunks]

    def run(by_blocks):
        model = ShapeModel(len(counts), Hq, Hkv, D, device=DEV)
        cache = DuoAttentionStaticKVCache(model, heads_from_counts(counts, Hkv), 1, sum(chunks) + 8, sink, recent)
        outs = []
        for ci, S in enumerate(chunks):
            pos = cache.kv_seq_len
            for li in range(len(counts)):
                q, k, v = (t.to(DEV) for t in data[ci][li])
                if by_blocks:
                    parts = [duo_static_attention_row_block(q[:, r0:r0 + rows], k[:, r0:r0 + rows], v[:, r0:r0 + rows],
                                                            cache, li, r0, S, 1.0, 1e4) for r0 in range(0, S, rows)]
                    outs.append(torch.cat(parts, 1))
                else:
                    outs.append(duo_static_attention_core(q, k, v, cache, li, pos, 1.0, 1e4))
        return outs, cache

    whole, c0 = run(False)
    blocked, c1 = run(True)
    for a, b in zip(whole, blocked):
        a, b = a.float().cpu(), b.float().cpu()
        assert torch.isfinite(b).all()
        assert (a - b).pow(2).mean().sqrt() <= 5e-3 * a.pow(2).mean().sqrt()
    assert c0.kv_seq_len_list == c1.kv_seq_len_list and c0.streaming_kv_seq_len_list == c1.streaming_kv_seq_len_list
    for l in range(len(counts)):
        n, m = c0.kv_seq_len_list[l], c0.streaming_kv_seq_len_list[l]
        for x, y, ln in ((c0.full_key_states_list, c1.full_key_states_list, n), (c0.full_value_states_list, c1.full_value_states_list, n),
                         (c0.streaming_key_states_list, c1.streaming_key_states_list, m),
                         (c0.streaming_value_states_list, c1.streaming_value_states_list, m)):
            assert torch.equal(x[l][:, :ln], y[l][:, :ln])


# ----------------------------------------------------------------------------- the fallback scan kernel stays tested
@pytest.mark.parametrize("case", [DECODE_CASES[2], DECODE_CASES[5], DECODE_CASES[8], DECODE_CASES[10], DECODE_CASES[12]])
def test_decode_on_the_long_prologue_scan_kernel(case):
    """duo_decode_split_kernel (round 2's scan) is what a launch falls back to when a stride does not fit the packed
    argument fields of duo_decode_scan_kernel; debug bit 9 selects it for any launch."""
    h = _hip()
    group, nf, ns, n_full, n_stream = case
    h.set_debug_flags(512)
    try:
        out, ref, bud = _attention_case(1, group, nf, ns, n_full, n_stream, True, seed=hash(case) % 1000 + 1)
    finally:
        h.set_debug_flags(0)
    attn_close(out, ref, f"decode (long-prologue kernel) {case}", bud)


def test_decode_falls_back_when_a_stride_does_not_fit_the_packed_fields():
    """q rows 70 000 elements apart (> 16 bits): the launcher must take the general kernel, with the same result as the
    short-prologue kernel on a compact copy of the same q."""
    from duo_attn.backend import HipBackend

    g = torch.Generator().manual_seed(77)
    group, nf, ns, N = 4, 2, 2, 3000
    Hq = (nf + ns) * group
    wide = torch.zeros(Hq, 70000, dtype=torch.bfloat16, device=DEV)
    wide[:, :D] = _rand((Hq, D), g).to(DEV)
    q_wide = wide[:, :D][None]                      # [1, Hq, D] view, head stride 70 000
    q_compact = q_wide.contiguous()
    fk, fv, fkd, fvd = _make_pool(N, nf, g, True)
    sk, sv, skd, svd = _make_pool(64, ns, g, True)
    kn, vn = _rand((1, nf + ns, D), g).to(DEV), _rand((1, nf + ns, D), g).to(DEV)
    full = (nf, 0, (fkd, fvd), (kn[:, :nf], vn[:, :nf]))
    stream = (ns, nf * group, (skd, svd), (kn[:, nf:], vn[:, nf:]))
    be = HipBackend()
    outs = []
    for q in (q_wide, q_compact):
        out = torch.full((1, Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        be.attention(q, out, group, full, stream, D ** -0.5)
        outs.append(out)
    assert torch.isfinite(outs[0].float()).all()
    # same partition, same arithmetic per workgroup up to the fold order of the epilogue (shared): equal to the bf16 ulp
    assert (outs[0].float() - outs[1].float()).abs().max() <= 2.0 ** -7 * outs[1].float().abs().max()
