"""GPU: INTEGRATION.md's reference-side ctypes stub is EXECUTED, not only parsed.

The first ```python block of INTEGRATION.md §B (``duo_attn/patch/hip_backend.py (new file on the reference side)``) is
extracted and exec'd as printed — only the library name is resolved to the in-tree ``libduoattn_hip.so`` — and its three
functions are called on device tensors: ``apply_rope_inplace`` (the replacement of flashinfer_utils.py:29-59) and
``duo_flash_attn`` in its prefill and decode forms (the replacement of the flash_attn_func calls of llama.py:364-421).
Results must equal, bit for bit, what this package's own binding (duo_attn/_hip.py) produces for the same inputs: the stub
declares no argtypes, so this is also the check that its bare Python ints / explicit c_int64 / c_float arguments reach the
C ABI intact."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 128


def _stub():
    from duo_attn import _hip

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = [b for b in re.findall(r"```python\n(.*?)```", doc, flags=re.S) if 'ctypes.CDLL("libduoattn_hip.so")' in b]
    assert len(block) == 1
    _hip.load_library()
    lib = os.path.join(ROOT, "duo-attention_amd", "lib", "libduoattn_hip.so")
    ns = {}
    exec(compile(block[0].replace('"libduoattn_hip.so"', repr(lib)), "INTEGRATION.md §B stub", "exec"), ns)
    return ns


def _rn(g, *shape):
    return torch.randn(*shape, generator=g).to(torch.bfloat16).to(DEV)


def _pool(g, rows, heads):      # head-major storage, token-major view, as DuoAttentionStaticKVCache allocates
    return _rn(g, heads, rows, D).permute(1, 0, 2)


def test_stub_rope_equals_the_package_binding():
    from duo_attn.patch.flashinfer_utils import apply_rope_inplace

    ns = _stub()
    g = torch.Generator().manual_seed(1)
    q, k = _rn(g, 2, 37, 8, D), _rn(g, 2, 37, 2, D)
    q2, k2 = q.clone(), k.clone()
    ns["apply_rope_inplace"](q, k, torch.tensor([1000, 1000]), 1.0, 500000.0)
    apply_rope_inplace(q2, k2, 1000, 1.0, 500000.0)
    torch.cuda.synchronize()
    assert torch.equal(q, q2) and torch.equal(k, k2) and not torch.equal(q, _rn(torch.Generator().manual_seed(1), 2, 37, 8, D))


@pytest.mark.parametrize("S,past", [(300, 900), (1, 5000), (64, 0)])
def test_stub_attention_equals_the_package_binding(S, past):
    from duo_attn import _hip
    from duo_attn.backend import HipBackend

    ns = _stub()
    g = torch.Generator().manual_seed(S + past)
    nf, n_s, G, W = 2, 2, 4, 96
    Hq, scale = (nf + n_s) * G, D ** -0.5
    q = _rn(g, S, Hq, D)
    kn, vn = _pool(g, S, nf + n_s), _pool(g, S, nf + n_s)
    if past == 0:       # first chunk: every head causal over the chunk (llama.py:364-372)
        full, stream = (nf + n_s, 0, None, (kn, vn)), None
    else:
        fk, fv = _pool(g, past + S, nf), _pool(g, past + S, nf)
        fk[past:], fv[past:] = kn[:, :nf], vn[:, :nf]
        sk, sv = _pool(g, W, n_s), _pool(g, W, n_s)
        full = (nf, 0, (fk[:past], fv[:past]), (fk[past:], fv[past:]))
        stream = (n_s, nf * G, (sk, sv), (kn[:, nf:], vn[:, nf:]))
    ws = torch.empty(_hip.load_library().duo_attn_decode_workspace_bytes(Hq, 512) // 4, dtype=torch.float32, device=DEV)
    a, b = torch.zeros_like(q), torch.zeros_like(q)
    ns["duo_flash_attn"](q, a, G, full, stream, scale, ws)
    # the stub binds duo_attn_prefill_bf16 (no workspace: one launch over the whole key range); the package's own call passes
    # a workspace and may split the key range of a small chunk over several launches — another summation order.  Debug bit 8
    # ("never split the key range") makes the package issue the stub's launch, so the comparison is bit for bit.
    _hip.set_debug_flags(256)
    try:
        HipBackend().attention(q, b, G, full, stream, scale)
    finally:
        _hip.set_debug_flags(0)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and a.float().abs().max() > 0
    assert torch.equal(a, b)


def test_stub_token_linear_equals_the_package_binding():
    """the ``token_linear`` wrapper of INTEGRATION.md §B "The token-row linears either side of it", exec'd as printed (up to
    its usage sketch) on top of the stub above: q|k|v with the RMSNorm prologue and a bias, and down_proj with the SiLU * up
    prologue and the residual epilogue — bit for bit what duo_attn/_hip.py's own wrapper returns"""
    from duo_attn import _hip

    ns = _stub()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = [b for b in re.findall(r"```python\n(.*?)```", doc, flags=re.S) if "def token_linear(" in b]
    assert len(block) == 1
    exec(compile(block[0].split("# decoder layer, q_len == 1")[0], "INTEGRATION.md token_linear stub", "exec"), ns)
    g = torch.Generator().manual_seed(7)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(DEV)
    h, nw = rn(2, 1024), (torch.rand(1024, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    blocks = [(rn(512, 1024, sc=1 / 32), rn(512)), (rn(128, 1024, sc=1 / 32), None), (rn(128, 1024, sc=1 / 32), None)]
    got = ns["token_linear"](h, blocks, norm=(nw, 1e-5))
    want = _hip.token_linear(h, blocks, norm=(nw, 1e-5))
    gu, res, wd = rn(2, 2 * 768), rn(2, 1024), rn(1024, 768, sc=1 / 28)
    got2 = ns["token_linear"](gu[:, :768], [(wd, None)], x2=gu[:, 768:], residual=res)
    want2 = _hip.token_linear(gu[:, :768], [(wd, None)], x2=gu[:, 768:], residual=res)
    got3 = ns["token_linear"](h, blocks[1:], norm=(nw, 1e-5), norm_hf=True)
    want3 = _hip.token_linear(h, blocks[1:], norm=(nw, 1e-5), norm_hf=True)
    torch.cuda.synchronize()
    for a, b in ((got, want), (got2, want2), (got3, want3)):
        assert a.shape == b.shape and torch.isfinite(a).all() and torch.equal(a, b)
