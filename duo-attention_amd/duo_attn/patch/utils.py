"""Head/weight reordering (host side, one-off).

Mirror of reference ``duo_attn/patch/utils.py:7-45``: permute the output rows of
q/k/v projections and the input columns of o_proj so that, per layer, retrieval
("full") heads come first and streaming heads last, each class keeping its
original relative order.  After this the kernels only need a split point
``nf`` per layer — never a gather.
"""
import torch


@torch.no_grad()
def reorder_linear_weights(linear_module: torch.nn.Linear, full_attention_heads: torch.Tensor,
                           repeat_num, reorder_channel):
    assert reorder_channel in ["in", "out"]
    mask = torch.repeat_interleave(full_attention_heads, repeats=repeat_num).to(linear_module.weight.device) > 0.5
    perm = torch.cat([torch.nonzero(mask).flatten(), torch.nonzero(~mask).flatten()])
    w = linear_module.weight.data
    if reorder_channel == "in":
        linear_module.weight.data = w.index_select(1, perm).contiguous()
    else:
        linear_module.weight.data = w.index_select(0, perm).contiguous()
        if linear_module.bias is not None:   # bias follows the output rows (the reference, patch/utils.py:27-32, also indexes
            #                                     the bias with the mask when the IN channels are reordered — an IndexError for a
            #                                     biased o_proj; there the bias stays, the output rows do not move)
            linear_module.bias.data = linear_module.bias.data.index_select(0, perm).contiguous()
    return linear_module


@torch.no_grad()
def reorder_full_attn_heads(full_attention_heads: torch.Tensor):
    n = int((full_attention_heads > 0.5).sum().item())
    full_attention_heads[:n] = 1
    full_attention_heads[n:] = 0
    return full_attention_heads
