"""INT4 KV pools against golden vectors recorded from the REFERENCE'S OWN KERNELS
(/root/reference/demo/quantize_int4.cu built for gfx950 by oracle/build_ref.py, run on an MI355X by
tests/golden/make_int4_golden.py -> tests/golden/int4_ref.npz).

CPU: the numpy oracle reproduces the `nocontract` build (the source as written) bit for bit, the `default`
build with ``fused=True``, and bounds its distance to the `fast` (approximate-reciprocal) build.
GPU: the product's duo_int4_quantize reproduces the `nocontract` build bit for bit, duo_int4_dequantize_f16 the `nocontract`
build (fused=False, the default) and the `default` build (fused=True); the fused decode kernels dequantise in the same form.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle.int4_oracle import dequantize_int4_ref, dequantize_int4_torch, quantize_int4_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "int4_ref.npz")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def case_names(g):
    return [k[2:] for k in g.files if k.startswith("x|")]


def f16(a):
    return a.view(np.float16)


# ----------------------------------------------------------------------------- CPU: oracle vs reference output
def test_oracle_quantize_equals_reference_source_build(g):
    for name in case_names(g):
        x = f16(g[f"x|{name}"])
        p, s, z = quantize_int4_ref(x)
        for variant in ("nocontract", "default"):    # both divide in IEEE fp32: identical codes
            assert np.array_equal(p, g[f"q|{name}|{variant}"]), (name, variant)
            assert np.array_equal(s.view(np.uint16), g[f"s|{name}|{variant}"]), (name, variant)
            assert np.array_equal(z.view(np.uint16), g[f"z|{name}|{variant}"]), (name, variant)


def test_oracle_dequantize_equals_reference_builds(g):
    for name in case_names(g):
        for variant, fused in (("nocontract", False), ("default", True), ("fast", True)):
            q, s, z = g[f"q|{name}|{variant}"], f16(g[f"s|{name}|{variant}"]), f16(g[f"z|{name}|{variant}"])
            got = dequantize_int4_ref(q, s, z, fused=fused)
            assert np.array_equal(got.view(np.uint16), g[f"dq|{name}|{variant}"]), (name, variant)
    for variant, fused in (("nocontract", False), ("default", True), ("fast", True)):
        got = dequantize_int4_ref(g["rawq"], f16(g["raws"]), f16(g["rawz"]), fused=fused)
        want = g[f"rawdq|{variant}"]
        same = (got.view(np.uint16) == want) | (np.isnan(got) & np.isnan(f16(want)))   # NaN payloads may differ
        assert same.all(), variant


def test_fused_and_unfused_dequant_really_differ(g):
    """the distinction is not academic: the contracted build rounds once"""
    a, b = g["rawdq|nocontract"], g["rawdq|default"]
    frac = (a != b).mean()
    assert 0.001 < frac < 0.5, frac


def test_fast_math_build_is_within_one_code_of_the_oracle(g):
    """-ffast-math turns (x - zero) / scale into x * v_rcp_f32(scale): codes may flip only where the exact
    quotient sits within a couple of fp32 ulps of k + 0.5.  Scale and zero point are unaffected."""
    report = {}
    for name in case_names(g):
        x = f16(g[f"x|{name}"]).astype(np.float32)
        p, s, z = quantize_int4_ref(f16(g[f"x|{name}"]))
        pf = g[f"q|{name}|fast"]
        assert np.array_equal(s.view(np.uint16), g[f"s|{name}|fast"]), name
        assert np.array_equal(z.view(np.uint16), g[f"z|{name}|fast"]), name
        hi_o, lo_o = (p >> 4).astype(np.int16), (p & 15).astype(np.int16)
        hi_f, lo_f = (pf >> 4).astype(np.int16), (pf & 15).astype(np.int16)
        codes_o = np.stack([hi_o, lo_o], -1).reshape(x.shape)
        codes_f = np.stack([hi_f, lo_f], -1).reshape(x.shape)
        d = codes_f - codes_o
        assert np.abs(d).max() <= 1, name
        # every flipped code is a near-tie of the exact quotient
        mn = x.min(-1, keepdims=True).astype(np.float64)
        mx = x.max(-1, keepdims=True).astype(np.float64)
        sc = ((mx - mn).astype(np.float32) / np.float32(15) + np.float32(1e-8)).astype(np.float64)
        quo = (x.astype(np.float64) - mn) / sc
        frac = np.abs(quo - np.floor(quo) - 0.5)
        assert (frac[d != 0] <= 1e-5 * np.maximum(quo[d != 0], 1)).all(), name
        report[name] = {"codes": int(d.size), "differ": int((d != 0).sum())}
    total = sum(r["differ"] for r in report.values())
    # random data essentially never lands on a tie; the engineered tie rows ("special") are where it shows
    for name in ("randn", "bf16vals", "small", "large", "offset"):      # fp16 inputs are discrete: a few exact ties exist
        assert report[name]["differ"] <= 1e-3 * report[name]["codes"], (name, report[name])
    print("fast-math build vs oracle (codes differing):", json.dumps(report), "total", total)


def test_torch_restatement_equals_numpy_oracle():
    """dequantize_int4_torch (used as the reference of the 3.3M-token cfg5 test, where numpy float16 would
    take minutes) is the same function as dequantize_int4_ref."""
    rng = np.random.default_rng(5)
    p = rng.integers(0, 256, (4096, 64), dtype=np.uint8)
    s = (rng.random(4096) * np.exp(rng.random(4096) * 12 - 9)).astype(np.float16)
    z = (rng.standard_normal(4096) * np.exp(rng.random(4096) * 8 - 4)).astype(np.float16)
    sz = torch.from_numpy(np.stack([s, z], -1))
    got = dequantize_int4_torch(torch.from_numpy(p), sz).numpy()
    assert np.array_equal(got.view(np.uint16), dequantize_int4_ref(p, s, z).view(np.uint16))


# ----------------------------------------------------------------------------- GPU: product vs reference output
gpu = pytest.mark.gpu


def _pools(T, h, head_major):
    if head_major:
        q = torch.zeros(h, T, 64, dtype=torch.uint8, device=DEV).permute(1, 0, 2)
        sz = torch.zeros(h, T, 2, dtype=torch.float16, device=DEV).permute(1, 0, 2)
    else:
        q = torch.zeros(T, h, 64, dtype=torch.uint8, device=DEV)
        sz = torch.zeros(T, h, 2, dtype=torch.float16, device=DEV)
    return q, sz


@gpu
@pytest.mark.parametrize("head_major", [True, False])
def test_hip_quantize_equals_reference_kernel(g, head_major):
    from duo_attn import _hip

    for name in case_names(g):
        x = torch.from_numpy(f16(g[f"x|{name}"]).copy())
        S, h, _ = x.shape
        row0 = 3
        q, sz = _pools(S + row0 + 1, h, head_major)
        _hip.int4_quantize(x.to(DEV), q, sz, row0)
        assert np.array_equal(q.cpu().numpy()[row0:row0 + S], g[f"q|{name}|nocontract"]), name
        got_sz = sz.cpu()[row0:row0 + S].numpy().view(np.uint16)
        assert np.array_equal(got_sz[..., 0], g[f"s|{name}|nocontract"]), name
        assert np.array_equal(got_sz[..., 1], g[f"z|{name}|nocontract"]), name
        if name == "bf16vals":     # the same values handed over as bf16 take the bf16 entry of the kernel
            q2, sz2 = _pools(S + row0 + 1, h, head_major)
            _hip.int4_quantize(x.to(torch.bfloat16).to(DEV), q2, sz2, row0)
            assert torch.equal(q2, q) and torch.equal(sz2, sz)


@gpu
@pytest.mark.parametrize("fused,variant", [(False, "nocontract"), (True, "default")])
def test_hip_dequantize_equals_reference_kernel(g, fused, variant):
    """both rounding forms of the product against the matching build of the reference's own kernel, bit for bit:
    hmul-then-hadd == the `nocontract` build (the source as written), fma == the `default` build (contracted)"""
    from duo_attn import _hip

    for name in case_names(g):
        qg = g[f"q|{name}|{variant}"]
        S, h, _ = qg.shape
        q, sz = _pools(S, h, True)
        q.copy_(torch.from_numpy(qg))
        sz[..., 0].copy_(torch.from_numpy(f16(g[f"s|{name}|{variant}"]).copy()))
        sz[..., 1].copy_(torch.from_numpy(f16(g[f"z|{name}|{variant}"]).copy()))
        out = torch.empty(S * h * 128, dtype=torch.float16, device=DEV)
        got = _hip.int4_dequantize(q, sz, S, out, fused=fused).cpu().numpy().view(np.uint16)
        assert np.array_equal(got, g[f"dq|{name}|{variant}"]), name
    N = g["rawq"].shape[0]
    q, sz = _pools(N, 1, False)
    q[:, 0].copy_(torch.from_numpy(g["rawq"]))
    sz[:, 0, 0].copy_(torch.from_numpy(f16(g["raws"]).copy()))
    sz[:, 0, 1].copy_(torch.from_numpy(f16(g["rawz"]).copy()))
    out = torch.empty(N * 128, dtype=torch.float16, device=DEV)
    got = _hip.int4_dequantize(q, sz, N, out, fused=fused).cpu().numpy().reshape(N, 128)
    want = f16(g[f"rawdq|{variant}"])
    same = (got.view(np.uint16) == want.view(np.uint16)) | (np.isnan(got) & np.isnan(want))
    assert same.all()


@gpu
@pytest.mark.parametrize("scalar_kernel", [False, True])
@pytest.mark.parametrize("fused", [False, True])
def test_hip_int4_decode_dequantises_in_the_requested_form(fused, scalar_kernel):
    """The fused decode attention must see exactly the values duo_int4_dequantize_f16 writes in the same form.  With a
    ONE-HOT softmax (one key scores far above the rest) the output is that key's dequantised V row, so the two rounding
    forms — which differ by one fp16 ulp on about half the values — are told apart in the attention's OUTPUT:
    out == dequantised V row (requested form) bit for bit, and != the other form's row."""
    from duo_attn import _hip

    gen = torch.Generator().manual_seed(3 + fused)
    T, h, G = 200, 2, 4
    kq, ksz = _pools(T, h, True)
    vq, vsz = _pools(T, h, True)
    k = torch.randn(T, h, 128, generator=gen).half() * 0.05
    v = (torch.randn(T, h, 128, generator=gen) * 3.0).half()
    hot = [57, 133]
    for j in range(h):
        k[hot[j], j] = torch.sign(torch.randn(128, generator=gen)).half() * 4.0          # |k_hot| = 4 in every dim
    _hip.int4_quantize(k.to(DEV), kq, ksz, 0)
    _hip.int4_quantize(v.to(DEV), vq, vsz, 0)
    scratch = torch.empty(T * h * 128, dtype=torch.float16, device=DEV)
    rows = {f: _hip.int4_dequantize(vq, vsz, T, scratch, fused=f).clone() for f in (False, True)}
    kd = _hip.int4_dequantize(kq, ksz, T, scratch, fused=fused).clone()
    q = torch.empty(h * G, 128, dtype=torch.float16, device=DEV)
    for j in range(h):
        q[j * G:(j + 1) * G] = kd[hot[j], j] * 8.0        # q.k_hot * scale = 128 * 16 * 8 / 11.3 >> any other score
    out = torch.empty_like(q)
    pool = _hip.make_int4_pool(kq, ksz, vq, vsz, T, 0)
    _hip.set_debug_flags(16 if scalar_kernel else 0)
    try:
        _hip.attn_decode_int4(q, out, G, pool, None, 128 ** -0.5, fused=fused)
    finally:
        _hip.set_debug_flags(0)
    differ = 0
    for j in range(h):
        want, other = rows[fused][hot[j], j], rows[not fused][hot[j], j]
        differ += int((want != other).sum())
        for gq in range(G):
            assert torch.equal(out[j * G + gq], want), (j, gq, (out[j * G + gq] != want).sum())
    assert differ > 0        # the fixture really distinguishes the two forms
