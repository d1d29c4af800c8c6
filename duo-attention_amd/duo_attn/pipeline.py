"""Layer pipeline across the GPUs of one node: one process per GPU, RCCL point-to-point.

The reference shards layers inside ONE process with accelerate hooks that copy hidden
states device-to-device (``duo_attn/utils.py:228-283``: even contiguous split of the decoder
layers, no collectives).  On MI355X the idiomatic form is one rank per GPU
(``torch.distributed`` backend "nccl" == RCCL) and the only exchange the path has: the hidden
state ``[B, S, hidden]`` crossing each stage boundary, sent point-to-point over a single xGMI
link (no all-reduce / all-to-all anywhere on the inference path).

Chunk-level pipelining is what makes prefill scale: chunk c on stage s depends only on chunk c
from stage s-1 and chunk c-1 on stage s, so with n chunks and P stages the makespan is
(n + P - 1) chunk-stage slots instead of n*P.  Sends are asynchronous (double-buffered) and the receive
of the next item is posted right behind the send of the current one, so a stage computes item i+1
while item i leaves.  Decode at batch 1 is strictly sequential across stages (latency = sum of stages + hops);
sharding it only multiplies KV capacity — this is reported as is.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .utils import balanced_layer_split, even_layer_split


class LayerPipeline:
    def __init__(self, num_layers: int, rank: Optional[int] = None, world_size: Optional[int] = None,
                 group=None, layer_costs: Optional[Sequence[float]] = None):
        """``layer_costs`` (one positive number per layer, e.g. ``base + n_full_kv_heads``) switches the
        reference's even split to the bottleneck-minimising contiguous split — the ragged per-layer
        retrieval-head counts otherwise leave every stage waiting for the heaviest one."""
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        if self.world_size > num_layers:
            raise ValueError(f"{self.world_size} stages for {num_layers} layers")
        if layer_costs is not None:
            if len(layer_costs) != num_layers:
                raise ValueError("layer_costs must have one entry per layer")
            self.bounds: List[Tuple[int, int]] = balanced_layer_split(layer_costs, self.world_size)
        else:
            self.bounds = even_layer_split(num_layers, self.world_size)
        self.first_layer, self.last_layer = self.bounds[self.rank]

    @property
    def is_first(self):
        return self.rank == 0

    @property
    def is_last(self):
        return self.rank == self.world_size - 1

    @property
    def layers(self):
        return range(self.first_layer, self.last_layer)

    def run(self, shapes: Sequence[Tuple[int, ...]], stage_fn: Callable[[int, Optional[torch.Tensor]], torch.Tensor],
            device, dtype=torch.bfloat16, token_feedback: Optional[Callable[[int, Optional[torch.Tensor]], torch.Tensor]] = None,
            ) -> List[Optional[torch.Tensor]]:
        """Stream ``len(shapes)`` items (prefill chunks or decode tokens) through this rank's stage.

        ``shapes[i]`` is the hand-off tensor shape of item i.  ``stage_fn(i, x)`` gets the hidden state
        received from the previous stage (``None`` on the first stage, which owns the inputs) and
        returns the hidden state for the next stage.  Returns the outputs of the LAST stage (a list
        of ``None`` elsewhere).

        ``token_feedback`` makes the stream autoregressive (batch-1 decode): item i+1 may not enter the
        first stage before item i has left the last one.  On the last stage it is called as
        ``token_feedback(i, y)`` and must return the int64 ``[B, 1]`` tensor to hand back (the sampled
        token); on the first stage it is called as ``token_feedback(i, token)`` with the tensor received
        for item i-1 -> i (its return value is ignored).  One extra 8-byte hop per item."""
        n = len(shapes)
        prev_rank, next_rank = self.rank - 1, self.rank + 1
        recv_bufs = [None, None]
        recv_work = [None, None]
        send_work = [None, None]
        keep_alive = [None, None]   # tensors of in-flight sends
        outs: List[Optional[torch.Tensor]] = []

        def post_recv(i):
            if self.is_first or i >= n:
                return
            slot = i & 1
            recv_bufs[slot] = torch.empty(shapes[i], device=device, dtype=dtype)
            recv_work[slot] = dist.irecv(recv_bufs[slot], src=prev_rank, group=self.group)

        feedback = token_feedback is not None and self.world_size > 1
        post_recv(0)
        for i in range(n):
            slot = i & 1
            x = None
            if feedback and self.is_first and i > 0:
                tok = torch.empty(shapes[i][0], 1, device=device, dtype=torch.int64)
                dist.recv(tok, src=self.world_size - 1, group=self.group)
                token_feedback(i, tok)
            if not self.is_first:
                recv_work[slot].wait()
                x = recv_bufs[slot]
            y = stage_fn(i, x)
            if self.is_last:
                outs.append(y)
                if feedback and i + 1 < n:
                    dist.send(token_feedback(i, y), dst=0, group=self.group)
            else:
                if send_work[slot] is not None:
                    send_work[slot].wait()   # the buffer of item i-2 has left
                keep_alive[slot] = y.contiguous()
                send_work[slot] = dist.isend(keep_alive[slot], dst=next_rank, group=self.group)
                outs.append(None)
            # The receive of item i+1 is posted AFTER the send of item i.  RCCL runs the point-to-point ops
            # a rank issues on one communicator in issue order: a receive posted ahead of the send would hold
            # the send back until the upstream stage has produced item i+1 — one extra item of latency per
            # stage while the pipeline fills — and, with token feedback, would wait on a token that needs
            # that very send (deadlock).  The hand-off (16 MB for a 2048-row block, ~0.1 ms on one xGMI
            # link) is small against an item's compute, so not overlapping it costs little.
            post_recv(i + 1)
        for w in send_work:
            if w is not None:
                w.wait()
        return outs
