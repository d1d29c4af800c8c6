"""GPU: the tuple-cache decode step in its fused form.

(1) ``duo_tuple_decode_prep_bf16`` against the oracle's ``tuple_decode_prep_ref`` (the reference's own torch sequence,
    llama.py:177-184, :202-223, :273-301): rotated q / k, arena rows, the new streaming cache — bit for bit, over the
    window cases (growing, exactly full, sliding, no sink / no recent, a class absent).
(2) the HuggingFace-form RMSNorm prologue of ``duo_token_linear_bf16`` (DUO_LINEAR_NORM_HF) against the oracle.
(3) whole HF models through ``enable_duo_attention_eval``: fused decode steps against the module-by-module tuple forward
    on the same GPU — every attention call replayed against the oracle, greedy tokens equal, caches equal up to the
    projections' summation order — and the path really is the fused one (launch counts).
"""
import copy

import numpy as np
import pytest
import torch

from oracle.duo_oracle import token_linear_ref, tuple_decode_prep_ref
from test_golden_and_model_gpu import _rel, tiny
from test_tuple_path_gpu import Recorder, check_calls_against_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D_ = 128


@pytest.mark.parametrize("nf,ns,n,sink,recent", [
    (1, 1, 5, 16, 48), (1, 1, 62, 16, 48), (1, 1, 63, 16, 48), (1, 1, 64, 16, 48), (2, 6, 384, 128, 256),
    (0, 2, 64, 16, 48), (2, 0, 7, 16, 48), (1, 1, 0, 16, 48), (1, 3, 100, 16, 48), (1, 1, 16, 16, 0), (3, 5, 200, 0, 64),
    (4, 4, 383, 128, 256)])
def test_tuple_decode_prep_matches_the_reference_sequence(nf, ns, n, sink, recent):
    from duo_attn import _hip

    g = torch.Generator().manual_seed(nf * 1000 + ns * 100 + n)
    rn = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16)
    Hkv, G, D, N = nf + ns, 4, 128, 37
    qkv = rn((Hkv * G + 2 * Hkv) * D)                     # one buffer, like the q|k|v projection's output
    ang = torch.rand(D // 2, generator=g) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()]).to(torch.bfloat16), torch.cat([ang.sin(), ang.sin()]).to(torch.bfloat16)
    arena = torch.zeros(2, nf, N + 9, D, dtype=torch.bfloat16)
    arena[:, :, :N] = rn(2, nf, N, D)
    # the old streaming cache in the layout the GENERAL tuple forward hands back: a transposed view of a token-major tensor
    old = rn(2, n, ns, D).transpose(1, 2)
    split = lambda t: (t[: Hkv * G * D].view(Hkv * G, D), t[Hkv * G * D:(Hkv * G + Hkv) * D].view(Hkv, D),
                       t[(Hkv * G + Hkv) * D:].view(Hkv, D))
    ref_buf, ref_arena = qkv.clone(), arena.clone()
    want = tuple_decode_prep_ref(*split(ref_buf), cos, sin, nf, ref_arena, N, old, sink, recent)
    dev_buf, dev_arena = qkv.to(DEV), arena.to(DEV)
    got = _hip.tuple_decode_prep(*split(dev_buf), cos.to(DEV), sin.to(DEV), nf, dev_arena, N, old.to(DEV), sink, recent)
    torch.cuda.synchronize()
    assert torch.equal(dev_buf.cpu(), ref_buf), "rotated q / k (and untouched v) differ"
    assert torch.equal(dev_arena.cpu(), ref_arena), "arena differs"
    assert got.shape == want.shape == (2, ns, min(n + 1, sink + recent), D) and got.is_contiguous()
    assert torch.equal(got.cpu(), want), "new streaming cache differs"


def test_tuple_decode_prep_refuses_what_it_cannot_do():
    from duo_attn import _hip

    lib = _hip.load_library()
    a = _hip.TupleDecodeArgs()
    assert lib.duo_tuple_decode_prep_bf16(None, None, None) == -1
    a.head_dim = 64
    assert lib.duo_tuple_decode_prep_bf16(a, None, None) == -2         # DUO_EHEADDIM
    a.head_dim = 128
    assert lib.duo_tuple_decode_prep_bf16(a, None, None) == -1         # null q


@pytest.mark.parametrize("rows,n_in", [(1, 512), (1, 4096), (3, 1024)])
def test_token_linear_hf_norm_prologue(rows, n_in):
    """DUO_LINEAR_NORM_HF: the normalised x is rounded to bf16 before the weight multiply (HF LlamaRMSNorm.forward), the
    bar of the other token-linear cases (one bf16 ulp of the exact product, >= 97 % bit-equal to the oracle)"""
    from duo_attn import _hip

    g = torch.Generator().manual_seed(rows * 7 + n_in)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16)
    x, nw = rn(rows, n_in, sc=3.0), (torch.rand(n_in, generator=g) + 0.5).to(torch.bfloat16)
    blocks = [(rn(384, n_in, sc=n_in ** -0.5), None), (rn(128, n_in, sc=n_in ** -0.5), rn(128))]
    dev = lambda t: None if t is None else t.to(DEV)
    for hf in (True, False):
        got = _hip.token_linear(dev(x), [(dev(w), dev(b)) for w, b in blocks], norm=(dev(nw), 1e-5), norm_hf=hf).cpu()
        want, pre = token_linear_ref(x, blocks, norm=(nw, 1e-5), exact=True, norm_hf=hf)
        err = (got.float() - pre.float()).abs()
        assert (err <= 2.0 ** -7 * pre.float().abs() + 1e-4).all(), (hf, err.max())
        assert (got == want).float().mean() >= 0.97, (hf, (got == want).float().mean())
    # the two forms are different functions: the identity-weight product exposes the extra rounding
    eye = torch.eye(n_in).to(torch.bfloat16)
    a = _hip.token_linear(dev(x), [(dev(eye), None)], norm=(dev(nw), 1e-5), norm_hf=True).cpu()
    b = _hip.token_linear(dev(x), [(dev(eye), None)], norm=(dev(nw), 1e-5), norm_hf=False).cpu()
    assert not torch.equal(a, b)
    from oracle.duo_oracle import rmsnorm_hf_ref

    assert (a == rmsnorm_hf_ref(x, nw, 1e-5)).float().mean() >= 0.98


class Counting(Recorder):
    def __init__(self, inner):
        super().__init__(inner)
        self.n_linear = self.n_prep = 0

    def token_linear(self, *a, **kw):
        self.n_linear += 1
        return self.inner.token_linear(*a, **kw)

    def tuple_decode_prep(self, *a, **kw):
        self.n_prep += 1
        return self.inner.tuple_decode_prep(*a, **kw)


@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_fused_tuple_decode_equals_module_by_module_on_the_gpu(family):
    from duo_attn import backend
    from duo_attn.patch import _duo, enable_duo_attention_eval

    base = tiny(family, seed=11)
    heads = np.array([[0.0, 1.0], [1.0, 1.0], [0.0, 0.0]])
    sink, recent = 16, 48                                   # window 64: full and sliding within the decode steps below
    ids = torch.randint(0, 211, (1, 90), generator=torch.Generator().manual_seed(12)).to(DEV)
    n_pre, n_dec = 58, 32

    def run(fused):
        model = copy.deepcopy(base)
        enable_duo_attention_eval(model, heads.copy(), sink, recent)
        rec = Counting(backend.HipBackend())
        backend._set_backend_for_testing(rec)
        old = _duo._FUSED_DECODE_LAYER
        _duo._FUSED_DECODE_LAYER = fused
        try:
            logits, toks = [], []
            with torch.no_grad():
                o = model(input_ids=ids[:, :n_pre], past_key_values=None, use_cache=True)
                past = o.past_key_values
                for t in range(n_pre, n_pre + n_dec):
                    o = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
                    past = o.past_key_values
                    logits.append(o.logits.float().cpu())
                    toks.append(int(o.logits[0, -1].argmax()))
        finally:
            _duo._FUSED_DECODE_LAYER = old
            backend._set_backend_for_testing(None)
        return torch.cat(logits, 1), toks, past, rec

    l_m, t_m, p_m, r_m = run(False)
    l_f, t_f, p_f, r_f = run(True)
    assert r_m.n_linear == 0 and r_m.n_prep == 0
    assert r_f.n_prep == n_dec * 3 and r_f.n_linear == n_dec * 3 * 4      # one prep + four token-row linears per layer, step
    check_calls_against_oracle(r_f.calls[3:], f"{family} fused tuple decode")        # (the 3 prefill calls: covered elsewhere)
    assert _rel(l_f, l_m) < 1e-2, _rel(l_f, l_m)
    assert sum(a == b for a, b in zip(t_f, t_m)) >= n_dec - 1
    for l in range(3):
        nf = int(heads[l].sum())
        assert p_f[l][0].shape == p_m[l][0].shape == (2, nf, n_pre + n_dec, 128)
        assert p_f[l][1].shape == p_m[l][1].shape == (2, 2 - nf, sink + recent, 128)
        for a, b in zip(p_f[l], p_m[l]):
            if b.numel():
                assert _rel(a, b) < 1e-2
    # prefill rows of layer 0 went through the same kernels in both runs
    assert torch.equal(p_f[0][0][:, :, :n_pre], p_m[0][0][:, :, :n_pre])


def test_fused_tuple_decode_long_context_llama3_geometry():
    """Llama-3-8B head geometry (32 q / 8 kv heads), one layer, 20 000-token context in the tuple format handed in as plain
    tensors (a foreign past: copied into the arena on the first step), three fused decode steps: every attention call
    against the oracle, the arena grows in place afterwards"""
    from transformers import LlamaConfig, LlamaForCausalLM

    from duo_attn import backend
    from duo_attn.patch import enable_duo_attention_eval

    torch.manual_seed(3)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=1024, num_hidden_layers=1, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=128, max_position_embeddings=1048576, rope_theta=3580165449.0,
                      attn_implementation="eager", tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval().to(DEV)
    enable_duo_attention_eval(model, np.array([[1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]), 128, 256)
    N = 20000
    g = torch.Generator(device=DEV).manual_seed(5)
    past = ((torch.randn(2, 3, N, 128, generator=g, device=DEV)).to(torch.bfloat16),
            (torch.randn(2, 5, 384, 128, generator=g, device=DEV)).to(torch.bfloat16))
    rec = Recorder(backend.HipBackend())
    backend._set_backend_for_testing(rec)
    try:
        pkv = (past,)
        with torch.no_grad():
            for t in range(3):
                o = model(input_ids=torch.tensor([[t + 1]], device=DEV), past_key_values=pkv, use_cache=True)
                pkv = o.past_key_values
                if t == 0:
                    first_ptr = pkv[0][0].data_ptr()
    finally:
        backend._set_backend_for_testing(None)
    assert pkv[0][0].shape == (2, 3, N + 3, 128) and pkv[0][1].shape == (2, 5, 384, 128)
    assert pkv[0][0].data_ptr() == first_ptr                      # appended in place after the first (copying) step
    assert torch.equal(pkv[0][0][:, :, :N], past[0])
    assert torch.equal(pkv[0][1][:, :, :128], past[1][:, :, :128])           # sink rows never move
    assert torch.equal(pkv[0][1][:, :, 128:381], past[1][:, :, 131:384])     # recent rows slid by three
    check_calls_against_oracle(rec.calls, "llama-3 geometry fused tuple decode")
    assert torch.isfinite(o.logits).all()


@pytest.mark.parametrize("nf,ns,N,n", [(3, 5, 5000, 384), (0, 2, 77, 64), (2, 0, 300, 0), (1, 1, 1, 1), (4, 4, 40000, 383)])
def test_stride_described_attention_equals_the_view_form(nf, ns, N, n):
    """``tuple_decode_attention`` (segments as pointers + strides of the tuple-format tensors) == the same launch described
    with tensor views through ``attention``: bit for bit, and within the attention bar of the oracle"""
    from helpers import attn_close
    from oracle.duo_oracle import flash_attn_func_ref

    from duo_attn.backend import HipBackend
    from duo_attn.patch._duo import tuple_decode_attention_by_views

    g = torch.Generator().manual_seed(N + n)
    rn = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(DEV)
    Hkv, G = nf + ns, 4
    arena, old = rn(2, nf, N + 50, D_), rn(2, ns, n, D_)
    qkv = rn((Hkv * G + 2 * Hkv), D_)
    q, k, v = qkv[: Hkv * G], qkv[Hkv * G: Hkv * G + Hkv], qkv[Hkv * G + Hkv:]
    be = HipBackend()
    a, b = torch.zeros_like(q), torch.zeros_like(q)
    be.tuple_decode_attention(q, a, G, nf, arena, N, old, k, v, D_ ** -0.5)
    tuple_decode_attention_by_views(be, q, b, G, nf, arena, N, old, k, v, D_ ** -0.5)
    assert torch.equal(a, b)
    c = lambda t: t.float().cpu()
    if nf:
        kk = torch.cat([c(arena[0, :, :N]).transpose(0, 1), c(k[:nf])[None]], 0).to(torch.bfloat16)
        vv = torch.cat([c(arena[1, :, :N]).transpose(0, 1), c(v[:nf])[None]], 0).to(torch.bfloat16)
        exact = flash_attn_func_ref(q[None, None, : nf * G].cpu(), kk[None], vv[None], causal=True, round_p=False,
                                    out_dtype=torch.float32)
        attn_close(a[None, None, : nf * G], exact, f"tuple stride form, retrieval nf={nf} N={N}")
    if ns:
        kk = torch.cat([c(old[0]).transpose(0, 1), c(k[nf:])[None]], 0).to(torch.bfloat16)
        vv = torch.cat([c(old[1]).transpose(0, 1), c(v[nf:])[None]], 0).to(torch.bfloat16)
        exact = flash_attn_func_ref(q[None, None, nf * G:].cpu(), kk[None], vv[None], causal=True, round_p=False,
                                    out_dtype=torch.float32)
        attn_close(a[None, None, nf * G:], exact, f"tuple stride form, streaming ns={ns} n={n}")


@pytest.mark.parametrize("S,Hq,Hkv", [(1, 8, 2), (67, 32, 8), (513, 4, 4)])
def test_hf_rotary_chunk_kernel_is_bit_equal_to_the_torch_sequence(S, Hq, Hkv):
    """duo_rope_hf_inplace_bf16 == transformers' apply_rotary_pos_emb(unsqueeze_dim=2) in bf16 (what the reference's tuple
    forward calls, llama.py:177-184), on the projections' layout and on a strided k view"""
    from duo_attn import _hip
    from duo_attn.patch.tuple_kv_cache import hf_apply_rotary_pos_emb

    g = torch.Generator().manual_seed(S + Hq)
    rn = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16)
    q, kbuf = rn(1, S, Hq, D_), rn(1, S, Hkv + 1, D_)
    k = kbuf[:, :, 1:]                                            # a view with a head offset (token stride > Hkv * D)
    pos = torch.arange(1000, 1000 + S)[:, None].float()
    inv = 1.0 / (500000.0 ** (torch.arange(0, D_, 2).float() / D_))
    ang = torch.cat([pos * inv, pos * inv], -1)
    cos, sin = ang.cos().to(torch.bfloat16)[None], ang.sin().to(torch.bfloat16)[None]
    wq, wk = hf_apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=2)                      # torch on the CPU
    dq, dkbuf = q.to(DEV), kbuf.to(DEV)
    dk = dkbuf[:, :, 1:]
    gq, gk = hf_apply_rotary_pos_emb(dq, dk, cos.to(DEV), sin.to(DEV), unsqueeze_dim=2)     # torch on the GPU
    _hip.rope_hf_inplace(dq[0], dk[0], cos[0].to(DEV), sin[0].to(DEV))
    assert torch.equal(dq.cpu(), wq) and torch.equal(dk.cpu(), wk)
    assert torch.equal(dq, gq) and torch.equal(dk, gk)
    assert torch.equal(dkbuf[:, :, 0].cpu(), kbuf[:, :, 0])                               # the neighbouring head is untouched


@pytest.mark.parametrize("rows,hidden", [(1, 512), (33, 4096), (7, 1000)])
def test_hf_rmsnorm_kernel_matches_the_transformers_module(rows, hidden):
    from transformers.models.llama.modeling_llama import LlamaRMSNorm

    from duo_attn import _hip
    from oracle.duo_oracle import rmsnorm_hf_ref

    g = torch.Generator().manual_seed(rows + hidden)
    x = (torch.randn(rows, hidden, generator=g) * 2.5).to(torch.bfloat16)
    norm = LlamaRMSNorm(hidden, eps=1e-5).to(torch.bfloat16)
    norm.weight.data = (torch.rand(hidden, generator=g) + 0.5).to(torch.bfloat16)
    want = norm(x)
    assert torch.equal(rmsnorm_hf_ref(x, norm.weight.data, 1e-5), want)
    got = _hip.rmsnorm_hf(x.to(DEV), norm.weight.data.to(DEV), 1e-5).cpu()
    # the statistics are summed in another order than torch's: a last-bit difference of rsqrt can move a rounding
    assert (got == want).float().mean() >= 0.995, (got == want).float().mean()
    assert ((got.float() - want.float()).abs() <= 2.0 ** -7 * want.float().abs() + 1e-30).all()
    on_gpu = norm.to(DEV)(x.to(DEV)).cpu()
    assert (got == on_gpu).float().mean() >= 0.995


def test_gradients_enabled_keep_the_tuple_decode_on_the_module_path():
    """the fused tuple step runs ctypes kernels that build no autograd graph, so a q_len == 1 step taken with
    gradients enabled on a model whose parameters require grad goes module by module (as tuple_rotary / _hf_norm already
    decide for themselves); under no_grad the same call is the fused one"""
    from duo_attn import backend
    from duo_attn.patch import enable_duo_attention_eval

    model = tiny("llama", seed=13)
    enable_duo_attention_eval(model, np.array([[0.0, 1.0], [1.0, 1.0], [0.0, 0.0]]), 16, 48)
    assert model.model.layers[0].self_attn.q_proj.weight.requires_grad
    ids = torch.randint(0, 211, (1, 40), generator=torch.Generator().manual_seed(14)).to(DEV)
    rec = Counting(backend.HipBackend())
    backend._set_backend_for_testing(rec)
    try:
        with torch.no_grad():
            past = model(input_ids=ids[:, :30], past_key_values=None, use_cache=True).past_key_values
        with torch.enable_grad():
            o = model(input_ids=ids[:, 30:31], past_key_values=past, use_cache=True)
        assert rec.n_prep == 0 and rec.n_linear == 0
        with torch.no_grad():
            o2 = model(input_ids=ids[:, 30:31], past_key_values=past, use_cache=True)
        assert rec.n_prep == 3 and rec.n_linear == 12
    finally:
        backend._set_backend_for_testing(None)
    assert _rel(o2.logits, o.logits.detach()) < 1e-2


def test_tuple_rotary_falls_back_on_cos_sin_rows_the_kernel_cannot_take():
    """cos / sin that are contiguous but not 16-byte aligned (a slice of a cached table at an odd offset) used to
    reach duo_rope_hf_inplace_bf16, which refuses them (DUO_EINVAL -> DuoHipError); now the torch sequence runs — same values"""
    from duo_attn.patch.tuple_kv_cache import hf_apply_rotary_pos_emb, tuple_rotary

    g = torch.Generator().manual_seed(15)
    rn = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(DEV)
    q, k = rn(1, 5, 4, D_), rn(1, 5, 2, D_)
    table = rn(2, 1 + 5 * D_)
    cos, sin = table[0, 1:].view(1, 5, D_), table[1, 1:].view(1, 5, D_)       # contiguous, data_ptr % 16 == 2
    assert cos.is_contiguous() and cos.data_ptr() % 16 != 0
    wq, wk = hf_apply_rotary_pos_emb(q.clone(), k.clone(), cos, sin, unsqueeze_dim=2)
    gq, gk = tuple_rotary(q.clone(), k.clone(), cos, sin)
    assert torch.equal(gq, wq) and torch.equal(gk, wk)
    # aligned rows of the same values: the kernel path, bit-equal to the sequence
    aq, ak = tuple_rotary(q.clone(), k.clone(), cos.clone(), sin.clone())
    assert torch.equal(aq, wq) and torch.equal(ak, wk)
