"""RoPE / RMSNorm entry points of the static path.

Same names and call shapes as reference ``duo_attn/patch/flashinfer_utils.py``
(``apply_rope_inplace`` :29-59, ``enable_flashinfer_rmsnorm`` :9-26), but the
arithmetic is this repo's HIP kernels (``duo_rope_inplace_bf16`` /
``duo_rmsnorm_bf16``) instead of the flashinfer CUDA package.
"""
import types

import torch

from ..backend import get_backend


def rmsnorm_forward(self, hidden_states):
    bsz, seq_len, hidden_size = hidden_states.size()
    out = get_backend().rmsnorm(hidden_states.reshape(bsz * seq_len, hidden_size), self.weight,
                                self.variance_epsilon)
    return out.view(bsz, seq_len, hidden_size)


flashinfer_rmsnorm_forward = rmsnorm_forward


def enable_flashinfer_rmsnorm(model):
    """Swap every *RMSNorm.forward for the fused HIP kernel (reference :19-26)."""
    for _, module in model.named_modules():
        if type(module).__name__ in ("LlamaRMSNorm", "MistralRMSNorm"):
            module.forward = types.MethodType(rmsnorm_forward, module)
    return model


def apply_rope_inplace(q: torch.Tensor, k: torch.Tensor, offsets, rope_scale: float, rope_theta: float,
                       indptr=None):
    """Rotate-half RoPE in place on q [B,S,Hq,D] and k [B,S,Hkv,D]; position = offset[b] + row.

    ``offsets`` may be a python int (no device read-back), a per-row list of ints, or a tensor like the
    reference passes (``position_ids[:, 0]``, one read-back)."""
    bsz = q.shape[0]
    if isinstance(offsets, torch.Tensor):
        offs = [int(o) for o in (offsets.expand(bsz) if offsets.numel() == 1 else offsets).tolist()]
    elif isinstance(offsets, (list, tuple)):
        offs = [int(o) for o in offsets]
        if len(offs) != bsz:
            raise ValueError(f"{len(offs)} RoPE offsets for a batch of {bsz}")
    else:
        offs = [int(offsets)] * bsz
    be = get_backend()
    if bsz > 1 and hasattr(be, "rope_inplace_batched"):
        be.rope_inplace_batched(q, k, offs, rope_scale, rope_theta)     # one launch when the rows share their position
        return q, k
    for b in range(bsz):
        be.rope_inplace(q[b], k[b], offs[b], rope_scale, rope_theta)
    return q, k
