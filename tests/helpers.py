"""Shared test helpers (shape-only model stub, tolerance checks)."""
import types

import torch


class ShapeModel:
    """What DuoAttentionStaticKVCache reads from a model: config dims + device/dtype of parameters."""

    def __init__(self, num_layers, num_heads, num_kv_heads, head_dim=128, device="cpu", dtype=torch.bfloat16):
        self.config = types.SimpleNamespace(
            num_hidden_layers=num_layers,
            num_attention_heads=num_heads,
            num_key_value_heads=num_kv_heads,
            hidden_size=num_heads * head_dim,
        )
        self._p = torch.zeros(1, device=device, dtype=dtype)

    def parameters(self):
        yield self._p


def heads_from_counts(counts, num_kv_heads):
    """per-layer [1]*nf + [0]*ns rows (already reordered form)."""
    return [[1.0] * nf + [0.0] * (num_kv_heads - nf) for nf in counts]


def attn_close(ours: torch.Tensor, ref_fp32: torch.Tensor, what=""):
    """Parity bar for bf16 attention outputs against the fp32 oracle.

    north_star asks for 1e-3 relative.  A bf16 OUTPUT cannot meet 1e-3 elementwise against an fp32
    value (bf16 half-ulp = 2^-9 = 1.95e-3 relative), so the check is
        |ours - ref| <= 1e-3*|ref|  +  2^-8*|ref| (one bf16 ulp of the value)  +  1e-3*rms(ref)
    (last term: absolute floor for elements that are sums cancelling to ~0), plus the
    flash-attn-test style bound  max|err| <= 2 * max|bf16(ref) - ref| + 1e-5.
    """
    o = ours.float().cpu()
    r = ref_fp32.float().cpu()
    assert o.shape == r.shape, (o.shape, r.shape)
    assert torch.isfinite(o).all(), f"{what}: non-finite output"
    err = (o - r).abs()
    rms = r.pow(2).mean().sqrt()
    tol = 1e-3 * r.abs() + (2.0 ** -8) * r.abs() + 1e-3 * rms
    bad = err > tol
    assert not bad.any(), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err {err.max():.3e} "
        f"at ref {r.flatten()[err.argmax()]:.3e}, rms {rms:.3e}"
    )
    bf16_floor = (r.to(torch.bfloat16).float() - r).abs().max()
    assert err.max() <= 2 * bf16_floor + 1e-5, f"{what}: max err {err.max():.3e} vs bf16 floor {bf16_floor:.3e}"
