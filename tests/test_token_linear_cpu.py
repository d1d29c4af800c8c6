"""CPU: ``token_linear_ref`` — the oracle of the fused decode-step linears (csrc/duo_linear.hip) — is the module sequence
it claims to be: an HF Llama decoder layer's non-attention half (reference static_kv_cache.py:482-537: input_layernorm,
q/k/v_proj, o_proj + residual, post_attention_layernorm, LlamaMLP, residual) evaluated module by module in bf16."""
import torch

from oracle.duo_oracle import rmsnorm_ref, token_linear_ref


def _ulp_close(a, b, what):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = (2.0 ** -7) * b.abs() + 1e-4
    assert (err <= tol).all(), f"{what}: worst {err.max().item():.3e}"
    assert (a == b).float().mean() >= 0.9, what


def test_ref_is_the_hf_module_sequence():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaMLP, LlamaRMSNorm

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    mlp = LlamaMLP(cfg).to(torch.bfloat16)
    ln = LlamaRMSNorm(256, eps=1e-5).to(torch.bfloat16)
    ln.weight.data = (torch.rand(256) + 0.5).to(torch.bfloat16)
    q = torch.nn.Linear(256, 256, bias=True).to(torch.bfloat16)
    k = torch.nn.Linear(256, 128, bias=False).to(torch.bfloat16)
    x = torch.randn(3, 256).to(torch.bfloat16)
    with torch.no_grad():
        # norm (HF's own RMSNorm rounds x*rs to the input dtype BEFORE the weight multiply; the static path swaps in the
        # flashinfer form — one rounding — which is what the oracle and the kernel implement: flashinfer_utils.py:9-26)
        xn = rmsnorm_ref(x, ln.weight, 1e-5)
        want_qk = torch.cat([q(xn), k(xn)], -1)
        got_qk = token_linear_ref(x, [(q.weight, q.bias), (k.weight, None)], norm=(ln.weight, 1e-5))
        _ulp_close(got_qk, want_qk, "norm + q|k")
        # o_proj-like product with the residual add
        want = x + q(x)
        got = token_linear_ref(x, [(q.weight, q.bias)], residual=x)
        _ulp_close(got, want, "linear + residual")
        # MLP: gate|up, then down over silu(g) * u, + residual
        gu = token_linear_ref(x, [(mlp.gate_proj.weight, None), (mlp.up_proj.weight, None)])
        got = token_linear_ref(gu[:, :704], [(mlp.down_proj.weight, None)], x2=gu[:, 704:], residual=x)
        want = x + mlp(x)
        _ulp_close(got, want, "mlp + residual")


def test_exact_returns_the_unrounded_product():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, generator=g).to(torch.bfloat16)
    w = torch.randn(5, 64, generator=g).to(torch.bfloat16)
    y, pre = token_linear_ref(x, [(w, None)], exact=True)
    assert pre.dtype == torch.float64 and torch.equal(y, pre.float().to(torch.bfloat16))
    assert torch.allclose(pre, x.double() @ w.double().t())
