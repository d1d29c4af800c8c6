"""Head-parallel tensor parallelism for the DuoAttention path.

The reference's long-context evaluations shard the model with the third-party ``tensor_parallel``
package (``duo_attn/utils.py:206-227``): device d takes the d-th contiguous block of kv heads and the
retrieval/streaming reorder then happens inside each shard (``llama.py:601-693``).  With DuoAttention
that split is lopsided — a layer's retrieval heads (full KV, all of the decode traffic) can all land on
one device.  Here one process owns one GPU (``torch.distributed``, backend "nccl" == RCCL over xGMI) and

* kv heads are dealt to the ranks so that every rank holds ``Hkv / tp`` heads per layer and the
  RETRIEVAL heads are spread evenly — per layer ⌊nf/tp⌋ or ⌈nf/tp⌉ each, the extra ones going to the
  ranks with the fewest so far (``balanced_head_assignment``), so KV bytes and decode time balance across
  the whole model, not per layer;
* q/k/v projections are column-sliced by head and ``o_proj`` row-sliced; the MLP is sliced the usual
  Megatron way; the two row-parallel outputs per layer are summed with ONE all-reduce each of
  ``[B, S, hidden]`` — the only collectives on the path;
* each rank then runs the ordinary single-GPU DuoAttention path on its slice: its heads are already
  ordered retrieval-first, so the patch's weight reorder is the identity, and its static KV cache only
  holds its own heads.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import nn


def balanced_head_assignment(full_attention_heads, tp: int) -> List[List[List[int]]]:
    """``full_attention_heads``: [L, Hkv] pattern (> 0.5 = retrieval head).  Returns ``assign[l][r]`` = the
    kv head ids of rank r in layer l, retrieval heads first (original order inside each class)."""
    heads = np.asarray(full_attention_heads, dtype=float)
    L, Hkv = heads.shape
    if Hkv % tp != 0:
        raise ValueError(f"{Hkv} kv heads do not divide over {tp} ranks")
    per = Hkv // tp
    load = [0] * tp                      # retrieval heads held so far, across layers
    out: List[List[List[int]]] = []
    for l in range(L):
        R = [h for h in range(Hkv) if heads[l, h] > 0.5]
        S = [h for h in range(Hkv) if heads[l, h] <= 0.5]
        base, extra = divmod(len(R), tp)
        if base + (1 if extra else 0) > per:
            raise ValueError("more retrieval heads than slots")   # cannot happen: len(R) <= Hkv
        # the ranks with the smallest cumulative load take the `extra` additional retrieval heads
        order = sorted(range(tp), key=lambda r: (load[r], r))
        n_r = [base] * tp
        for r in order[:extra]:
            n_r[r] += 1
        ranks: List[List[int]] = []
        ri = si = 0
        for r in range(tp):
            mine = R[ri:ri + n_r[r]]
            ri += n_r[r]
            fill = per - len(mine)
            mine = mine + S[si:si + fill]
            si += fill
            load[r] += n_r[r]
            ranks.append(mine)
        out.append(ranks)
    return out


def _host_staged(group) -> bool:
    """gloo cannot reduce / gather device memory: a gloo group with GPU tensors (the one-GPU rehearsal of the
    multi-rank path — every rank computes on cuda:0 — tests/test_sharded_models_gpu.py) stages through the host.
    RCCL ("nccl") moves device memory directly."""
    return dist.get_backend(group) == "gloo"


def all_reduce_sum(y: torch.Tensor, group=None) -> torch.Tensor:
    """in-place sum over the TP group of a row-parallel output [B, S, hidden]"""
    if y.is_cuda and _host_staged(group):
        h = y.detach().to("cpu", torch.float32)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        y.copy_(h.to(y.dtype))
        return y
    dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
    return y


class RowParallelLinear(nn.Module):
    """y = all_reduce_sum(x_local @ W_local^T): the slice of a Linear over its INPUT features."""

    def __init__(self, linear: nn.Linear, in_index: torch.Tensor, group=None, add_bias: bool = True):
        super().__init__()
        self.group = group
        w = linear.weight.data[:, in_index].contiguous()
        self.inner = nn.Linear(w.shape[1], w.shape[0], bias=linear.bias is not None and add_bias,
                               device=w.device, dtype=w.dtype)
        self.inner.weight.data.copy_(w)
        if self.inner.bias is not None:
            self.inner.bias.data.copy_(linear.bias.data)

    @property
    def weight(self):
        return self.inner.weight

    @property
    def bias(self):
        return self.inner.bias

    def forward(self, x):
        return all_reduce_sum(self.inner(x), self.group)


def _slice_rows(linear: nn.Linear, rows: torch.Tensor) -> nn.Linear:
    w = linear.weight.data[rows].contiguous()
    new = nn.Linear(w.shape[1], w.shape[0], bias=linear.bias is not None, device=w.device, dtype=w.dtype)
    new.weight.data.copy_(w)
    if linear.bias is not None:
        new.bias.data.copy_(linear.bias.data[rows])
    return new


def shard_model_for_tp(model, full_attention_heads, rank: Optional[int] = None, tp: Optional[int] = None,
                       group=None) -> np.ndarray:
    """Slice a HuggingFace Llama/Mistral ``*ForCausalLM`` in place for this rank and return the rank's
    pattern ``[L, Hkv / tp]`` (retrieval heads first) to hand to ``enable_*_duo_attention*_eval`` and to the
    KV cache constructor.  Embeddings, norms and ``lm_head`` stay replicated."""
    rank = dist.get_rank(group) if rank is None else rank
    tp = dist.get_world_size(group) if tp is None else tp
    cfg = model.config
    Hq, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = getattr(cfg, "head_dim", None) or cfg.hidden_size // Hq
    G = Hq // Hkv
    heads = np.asarray(full_attention_heads, dtype=float)
    # A model that already went through a DuoAttention enabler has its kv heads REORDERED (retrieval heads first, per
    # layer; `_duo_head_order`: position -> original head id).  The split works on positions; `heads` is in the original
    # head order, so it is taken through each layer's order first.
    perms = [list(layer.self_attn.__dict__.get("_duo_head_order") or range(Hkv)) for layer in model.model.layers]
    cur = np.stack([heads[l][perms[l]] for l in range(len(perms))]) if len(perms) else heads
    assign = balanced_head_assignment(cur, tp)
    inter = cfg.intermediate_size
    if inter % tp != 0:
        raise ValueError(f"intermediate_size {inter} does not divide over {tp} ranks")
    local = np.zeros((len(assign), Hkv // tp))
    for l, layer in enumerate(model.model.layers):
        kv_ids = assign[l][rank]
        local[l] = [1.0 if cur[l, h] > 0.5 else 0.0 for h in kv_ids]
        attn = layer.self_attn
        dev = attn.q_proj.weight.device
        kv_rows = torch.tensor([h * D + d for h in kv_ids for d in range(D)], device=dev)
        q_rows = torch.tensor([(h * G + g) * D + d for h in kv_ids for g in range(G) for d in range(D)], device=dev)
        attn.q_proj = _slice_rows(attn.q_proj, q_rows)
        attn.k_proj = _slice_rows(attn.k_proj, kv_rows)
        attn.v_proj = _slice_rows(attn.v_proj, kv_rows)
        attn.o_proj = RowParallelLinear(attn.o_proj, q_rows, group, add_bias=rank == 0)
        for name in ("num_heads", "num_key_value_heads", "num_key_value_groups"):
            if hasattr(attn, name) and name != "num_key_value_groups":
                setattr(attn, name, getattr(attn, name) // tp)
        if "full_attention_heads" in attn._buffers:
            # tuple-path patch already applied (enable_duo_attention_eval): the registered pattern follows the heads
            buf = attn._buffers["full_attention_heads"]
            attn._buffers["full_attention_heads"] = buf[torch.tensor(kv_ids, device=buf.device)].clone()
            attn.full_attn_head_mask = None
        attn.__dict__["_duo_head_order"] = None     # (positions are the rank's local ones from here on)
        mlp = layer.mlp
        cols = torch.arange(rank * inter // tp, (rank + 1) * inter // tp, device=dev)
        mlp.gate_proj = _slice_rows(mlp.gate_proj, cols)
        mlp.up_proj = _slice_rows(mlp.up_proj, cols)
        mlp.down_proj = RowParallelLinear(mlp.down_proj, cols, group, add_bias=rank == 0)
    # every module shares this config object: the per-rank head counts are what the patched forwards read
    cfg.head_dim = D
    cfg.num_attention_heads = Hq // tp
    cfg.num_key_value_heads = Hkv // tp
    cfg.intermediate_size = inter // tp
    # what the TP-aware head accessors (duo_attn.patch.get/set/map_full_attention_heads) need.  `assign` indexes the head
    # order the model had WHEN IT WAS SHARDED (the original order, or an enabler's reordered one: `perm` maps those
    # positions back to original head ids)
    model._duo_tp = {"assign": assign, "rank": rank, "tp": tp, "group": group, "num_kv_heads": Hkv, "perm": perms}
    return local


def tp_local_rows(model, full_attention_heads):
    """Head-pattern rows for THIS rank of a model sharded with ``shard_model_for_tp`` / ``to_device(enable_tp=True)``: a
    whole-model pattern ``[L, Hkv]`` (original kv-head order) is sliced through the rank's head assignment — the rank's
    heads come retrieval-first BY THE PATTERN THE MODEL WAS SHARDED WITH; with another pattern the rows are simply the
    rank's heads in its local order (the enablers' weight reorder then sorts them).  Rows that already have the local
    width pass through."""
    info = getattr(model, "_duo_tp", None)
    if info is None:
        return full_attention_heads
    rows = [np.asarray(torch.as_tensor(r).float().cpu() if torch.is_tensor(r) else r, dtype=float).reshape(-1)
            for r in full_attention_heads]
    if not rows or rows[0].shape[0] != info["num_kv_heads"]:
        return full_attention_heads
    if len(rows) != len(info["assign"]):
        raise ValueError(f"{len(rows)} pattern rows for {len(info['assign'])} layers")
    perm = info.get("perm") or [list(range(info["num_kv_heads"]))] * len(rows)
    return np.stack([rows[l][[perm[l][c] for c in info["assign"][l][info["rank"]]]] for l in range(len(rows))])


def gather_full_attention_heads(model, local_heads):
    """``local_heads``: this rank's per-layer buffers ``[Hkv / tp]`` (rank-local order, retrieval heads first).
    Returns the whole model's per-layer ``[Hkv]`` tensors in the head order the model had when it was sharded (the original
    order; for a model sharded AFTER an enabler, the enabler's reordered one — exactly what the unsharded patched model's
    buffers hold), identical on every rank — what the reference's TP branch assembles by concatenating its shards
    (llama.py:601-620)."""
    info = model._duo_tp
    tp, group = info["tp"], info["group"]
    out = []
    for l, mine in enumerate(local_heads):
        src = mine.contiguous()
        if src.is_cuda and _host_staged(group):
            src = src.to("cpu", torch.float32)
        parts = [torch.empty_like(src) for _ in range(tp)]
        dist.all_gather(parts, src, group=group)
        full = torch.empty(info["num_kv_heads"], dtype=mine.dtype, device=mine.device)
        for r in range(tp):
            full[torch.tensor(info["assign"][l][r], device=mine.device)] = parts[r].to(mine.device, mine.dtype)
        out.append(full)
    return out


def scatter_full_attention_heads(model, layer_idx, full_row):
    """this rank's slice (rank-local order) of a whole-model ``[Hkv]`` row given in the original head order"""
    info = model._duo_tp
    ids = torch.tensor(info["assign"][layer_idx][info["rank"]], device=full_row.device)
    return full_row[ids]
