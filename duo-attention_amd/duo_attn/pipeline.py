"""Layer pipeline across the GPUs of one node: one process per GPU, RCCL point-to-point.

The reference shards layers inside ONE process with accelerate hooks that copy hidden
states device-to-device (``duo_attn/utils.py:228-283``: even contiguous split of the decoder
layers, no collectives).  On MI355X the idiomatic form is one rank per GPU
(``torch.distributed`` backend "nccl" == RCCL) and the only exchange the path has: the hidden
state ``[B, S, hidden]`` crossing each stage boundary, sent point-to-point over a single xGMI
link (no all-reduce / all-to-all anywhere on the inference path).

Chunk-level pipelining is what makes prefill scale: chunk c on stage s depends only on chunk c
from stage s-1 and chunk c-1 on stage s, so with n chunks and P stages the makespan is
(n + P - 1) chunk-stage slots instead of n*P.  Every hand-off of the stream is issued through ``batch_isend_irecv`` — so both
ends of every hop are on the group's own communicator whatever way the process group was initialised (see ``run``) — and on the
middle stages the send of item i and the receive of item i+1 are ONE group, so neither waits behind the other on the rank's
in-order communicator and a stage computes item i+1 while item i leaves.  (Where the backend hands back one request per
operation — gloo — the receive is waited for before computing and the send only before its buffer is reused.  RCCL returns ONE
request for the coalesced group, so there the wait before item i+1 also covers the send of item i: a stage cannot run more than
one item ahead of its successor — the depth-1 buffering the double-buffered slots give anyway.  This has only ever run over
gloo: no RCCL run exists)  Decode at batch 1
is strictly sequential across stages (latency = sum of stages + hops); sharding it only multiplies KV
capacity — this is reported as is.

``PipelinedCausalLM`` is the model-level entry point (reference ``to_device(model, devices, enable_pp=True)``,
``duo_attn/utils.py:228-283``): this rank keeps its contiguous block of decoder layers (embedding on the
first stage, final norm + lm_head on the last), owns the dual KV pools of those layers, and streams prefill
chunks — whole, or in row blocks for a finer wavefront — and autoregressive decode tokens through
``LayerPipeline``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .utils import balanced_layer_split, even_layer_split


class _WorkGroup:
    """the request list of one batch_isend_irecv call behind a single wait()"""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


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

    def peer(self, stage: int) -> int:
        """Stage index (= rank inside ``group``) -> the GLOBAL rank torch.distributed's point-to-point calls take as
        ``src`` / ``dst`` / ``P2POp`` peer.  Identity for the default group; a pipeline that lives on a sub-group
        (stages on ranks 4..7, one pipeline per tensor-parallel slice) needs the translation."""
        return dist.get_global_rank(self.group, stage) if self.group is not None else stage

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
        prev_rank = self.peer(self.rank - 1) if self.rank > 0 else -1
        next_rank = self.peer(self.rank + 1) if self.rank + 1 < self.world_size else -1
        first_rank, last_rank = self.peer(0), self.peer(self.world_size - 1)
        recv_bufs = [None, None]
        recv_work = [None, None]
        send_work = [None, None]
        keep_alive = [None, None]   # tensors of in-flight sends
        outs: List[Optional[torch.Tensor]] = []

        # EVERY point-to-point operation of the stream — singletons included — is issued through batch_isend_irecv.
        # Why: torch's RCCL/NCCL process group picks the COMMUNICATOR of a point-to-point call by how it is issued and by
        # how the group was initialised: inside a batch it is always the group's own communicator; a plain isend / irecv
        # runs on the group's communicator when the group was initialised eagerly (init_process_group(device_id=...)) but
        # on a separate two-rank communicator of its pair otherwise — and operations on different communicators never
        # match.  Rounds 2-4 paired send(i) with recv(i+1) in a batch on the middle stages while the first / last stage
        # and the decode stream issued plain calls: correct under eager initialisation (what bench.py uses), a hang on
        # three or more GPUs for a caller that initialised lazily (gloo, the only transport this had run on, matches by
        # source and tag alone and shows neither).  With every call batched both ends of every hop are on the group's
        # communicator in either mode.  That communicator runs a rank's operations IN ISSUE ORDER, which the order of the
        # calls below respects (a send completes against the receive posted opposite it; see the notes at each site).
        def p2p(*ops):
            return dist.batch_isend_irecv([dist.P2POp(fn, t, peer, group=self.group) for fn, t, peer in ops])

        def post_recv(i):
            if self.is_first or i >= n:
                return
            slot = i & 1
            recv_bufs[slot] = torch.empty(shapes[i], device=device, dtype=dtype)
            recv_work[slot] = _WorkGroup(p2p((dist.irecv, recv_bufs[slot], prev_rank)))

        feedback = token_feedback is not None and self.world_size > 1
        tok_out = None              # the token tensor of the last stage's in-flight hop back
        post_recv(0)
        for i in range(n):
            slot = i & 1
            x = None
            if feedback and self.is_first and i > 0:
                tok = torch.empty(shapes[i][0], 1, device=device, dtype=torch.int64)
                _WorkGroup(p2p((dist.irecv, tok, last_rank))).wait()
                token_feedback(i, tok)
            if not self.is_first:
                recv_work[slot].wait()
                x = recv_bufs[slot]
            y = stage_fn(i, x)
            if self.is_last:
                outs.append(y)
                if feedback and i + 1 < n:
                    tok_out = token_feedback(i, y).contiguous()
                    _WorkGroup(p2p((dist.isend, tok_out, first_rank))).wait()
                post_recv(i + 1)
                continue
            outs.append(None)
            if send_work[slot] is not None:
                send_work[slot].wait()   # the buffer of item i-2 has left
            keep_alive[slot] = y.contiguous()
            if feedback or self.is_first or i + 1 >= n:
                # Autoregressive stream: the receive of item i+1 must be issued AFTER the send of item i — the communicator
                # runs a rank's operations in issue order, and a receive posted first would wait on a token that needs
                # that very send (deadlock).  (First stage / last item: nothing to pair with.)
                send_work[slot] = _WorkGroup(p2p((dist.isend, keep_alive[slot], next_rank)))
                post_recv(i + 1)
            else:
                # Feed-forward stream (prefill): send(i) and recv(i+1) go out as ONE group, so the receive does not queue
                # behind the send (nor the send behind an earlier receive) on the in-order communicator: a stage computes
                # item i+1 while item i leaves.
                nslot = (i + 1) & 1
                recv_bufs[nslot] = torch.empty(shapes[i + 1], device=device, dtype=dtype)
                works = p2p((dist.isend, keep_alive[slot], next_rank), (dist.irecv, recv_bufs[nslot], prev_rank))
                if len(works) == 2:
                    # the backend hands back one request per operation (gloo; RCCL returns ONE for the coalesced group):
                    # wait for the receive before computing and for the send only before its buffer is reused, so a slow
                    # downstream stage does not hold up this stage's next item
                    send_work[slot], recv_work[nslot] = _WorkGroup(works[:1]), _WorkGroup(works[1:])
                else:
                    # one request for the pair: the wait before item i+1 also covers the send of item i — a stage cannot run
                    # more than one item ahead of its successor, the depth the double-buffered slots give anyway
                    send_work[slot] = recv_work[nslot] = _WorkGroup(works)
        for w in send_work:
            if w is not None:
                w.wait()
        return outs


# =============================================================================
# two layer blocks per rank (virtual stages)
# =============================================================================
def _interleaved_units(n: int, group_size: int):
    """a rank's work in order: (item, pass) — groups of ``group_size`` items, pass 0 of a group, then its pass 1"""
    out = []
    for g0 in range(0, n, group_size):
        items = range(g0, min(n, g0 + group_size))
        out += [(i, 0) for i in items] + [(i, 1) for i in items]
    return out


def simulate_interleaved(block_costs: Sequence[float], world_size: int, group_size: int, n_items: int) -> float:
    """makespan of ``n_items`` equal items through 2 * world_size blocks (rank r owns blocks r and world_size + r), every rank
    working through ``_interleaved_units`` in order, a unit starting when its rank is free and its input exists (hand-off
    time not counted) — the planning model ``interleaved_layer_split`` minimises, in units of ``block_costs``"""
    P = world_size
    seq = _interleaved_units(n_items, group_size)
    done, free, pos = {}, [0.0] * P, [0] * P
    progressed = True
    while progressed:
        progressed = False
        for rk in range(P):
            while pos[rk] < len(seq):
                i, ps = seq[pos[rk]]
                vs = ps * P + rk
                if vs > 0 and (vs - 1, i) not in done:
                    break
                done[(vs, i)] = free[rk] = max(free[rk], done[(vs - 1, i)] if vs > 0 else 0.0) + block_costs[vs]
                pos[rk] += 1
                progressed = True
    return done[(2 * P - 1, n_items - 1)]


def interleaved_layer_split(costs: Sequence[float], world_size: int, group_sizes: Optional[Sequence[int]] = None,
                            ) -> Tuple[List[Tuple[int, int]], int]:
    """2 * world_size contiguous layer blocks, rank r owning blocks r and world_size + r, and the group size of the schedule:
    ``(bounds, group_size)``, chosen to minimise the SIMULATED makespan of a stream of equal items (``simulate_interleaved``).

    Why simulated: 32 ragged layers cut into 8 contiguous pieces leave the idlest stage 0.61 busy (profiles/
    r5_scaling_model.md), and two blocks per rank give 15 cuts instead of 7 and let a heavy block be paired with a light one —
    but balancing the ranks' LOADS alone (heaviest rank 4 % above the mean on 8 ranks) is not enough: every rank works
    through the same fixed unit order, so blocks of very different size stall each other (a rank's pass 1 waits for the
    item to come round), and the load-balanced split measured WORSE than one block per rank in the model.  The search —
    single cuts by one or two layers, pairs of cuts, from several starting points, for group sizes P and 2P — is
    deterministic (every rank computes the same answer) and a few seconds of host time at 32 layers / 8 ranks."""
    n, P = len(costs), int(world_size)
    if 2 * P > n:
        raise ValueError(f"{2 * P} blocks for {n} layers")
    pre = [0.0]
    for c in costs:
        pre.append(pre[-1] + float(c))
    scale = pre[-1] if pre[-1] > 0 else 1.0
    n_sim = 6 * P

    def blocks_of(cuts):
        b = [0] + cuts + [n]
        return [(pre[b[i + 1]] - pre[b[i]]) / scale for i in range(2 * P)]

    def climb(cuts, gs):
        best = simulate_interleaved(blocks_of(cuts), P, gs, n_sim)
        improved = True
        while improved:
            improved = False
            m = len(cuts)
            moves = [((i, d),) for i in range(m) for d in (-2, -1, 1, 2)]
            moves += [((i, d), (j, e)) for i in range(m) for j in range(i + 1, min(m, i + 4)) for d in (-1, 1) for e in (-1, 1)]
            for mv in moves:
                trial = list(cuts)
                for i, d in mv:
                    trial[i] += d
                bb = [0] + trial + [n]
                if any(bb[t + 1] <= bb[t] for t in range(len(bb) - 1)):
                    continue
                v = simulate_interleaved(blocks_of(trial), P, gs, n_sim)
                if v < best - 1e-12:
                    cuts, best, improved = trial, v, True
                    break
        return cuts, best

    starts = [[hi for _, hi in balanced_layer_split(costs, 2 * P)][:-1], [hi for _, hi in even_layer_split(n, 2 * P)][:-1]]
    h = next(i for i in range(1, n) if pre[i] >= pre[-1] / 2)      # two balanced halves around the half-way point
    h = min(max(h, P), n - P)
    starts.append([hi for _, hi in balanced_layer_split(costs[:h], P)] + [h + hi for _, hi in balanced_layer_split(costs[h:], P)][:-1])
    best = None
    for gs in (group_sizes or (P, 2 * P)):
        for st in starts:
            c, v = climb(list(st), gs)
            if best is None or v < best[0] - 1e-12:
                best = (v, c, gs)
    b = [0] + best[1] + [n]
    return [(b[i], b[i + 1]) for i in range(2 * P)], best[2]


class InterleavedLayerPipeline:
    """The layer pipeline with TWO layer blocks per rank: an item passes rank 0, 1, ..., P-1 (blocks 0 ... P-1), returns to
    rank 0 and passes them all again (blocks P ... 2P-1).  Reference: the same contiguous layer sharding
    (``duo_attn/utils.py:251-271``), cut finer; this form exists for balance and fill, it changes no arithmetic.

    Schedule (every rank the same, so every link carries ONE ordered stream): items in groups of ``group_size`` (P or 2P,
    chosen with the split); for each group first pass 0 of its items, then pass 1 of the same items.  With equal blocks
    rank r is busy from tick r on with no gaps (pass 1 of an item arrives exactly when pass 0 of its group is done), so the
    fill is r HALF-size units instead of r whole stages; a layer block still sees its items in order (its KV pools are
    appended in order).  An autoregressive stream (batch-1
    decode, ``token_feedback``) is groups of one: pass 0, pass 1, token back to rank 0.

    Hand-off discipline = ``LayerPipeline.run``'s: every point-to-point operation goes through ``batch_isend_irecv``; the send
    of a unit and the receive of the NEXT unit are one batch (so neither queues behind the other on the in-order
    communicator), never earlier — on rank 0 a receive from rank P-1 posted before the sends it depends on would stop the
    ring.  Checked like the one-block pipeline: gloo runs equal to one process, ``helpers.P2PAudit`` replayed on an
    in-order communicator in both initialisation modes.  Never run on RCCL (no multi-GPU node): opt-in
    (``bench.py --virtual-stages 2``)."""

    def __init__(self, num_layers: int, rank: Optional[int] = None, world_size: Optional[int] = None, group=None,
                 layer_costs: Optional[Sequence[float]] = None, group_size: Optional[int] = None):
        """``group_size``: force the schedule's group size (a multiple of the ring length would be usual; any value >= 1 is
        served, shorter-than-ring groups with bubbles); None = chosen with the split"""
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        costs = list(layer_costs) if layer_costs is not None else [1.0] * num_layers
        if len(costs) != num_layers:
            raise ValueError("layer_costs must have one entry per layer")
        self.bounds, self.group_size = interleaved_layer_split(costs, self.world_size, [int(group_size)] if group_size else None)
        self.blocks = (self.bounds[self.rank], self.bounds[self.world_size + self.rank])      # (pass 0, pass 1)

    def peer(self, stage: int) -> int:
        return dist.get_global_rank(self.group, stage) if self.group is not None else stage

    @property
    def is_first(self):
        return self.rank == 0

    @property
    def is_last(self):
        return self.rank == self.world_size - 1

    def layers(self, pass_idx: int):
        lo, hi = self.blocks[pass_idx]
        return range(lo, hi)

    def units(self, n: int, group_size: int):
        return _interleaved_units(n, group_size)

    def run(self, shapes: Sequence[Tuple[int, ...]], stage_fn: Callable[[int, Optional[torch.Tensor], int], torch.Tensor],
            device, dtype=torch.bfloat16, token_feedback: Optional[Callable[[int, Optional[torch.Tensor]], torch.Tensor]] = None,
            ) -> List[Optional[torch.Tensor]]:
        """``stage_fn(i, x, pass_idx)``: item i through this rank's block of that pass; ``x`` is None for pass 0 on rank 0
        (which owns the inputs).  Returns the outputs of pass 1 on the LAST rank (a list of None elsewhere).
        ``token_feedback``: as ``LayerPipeline.run`` — item i+1 enters rank 0 only after item i left the last rank."""
        n, P = len(shapes), self.world_size
        if P == 1:
            outs = []
            for i in range(n):
                outs.append(stage_fn(i, stage_fn(i, None, 0), 1))
                if token_feedback is not None and i + 1 < n:
                    token_feedback(i + 1, token_feedback(i, outs[-1]))
            return outs
        feedback = token_feedback is not None
        seq = self.units(n, 1 if feedback else self.group_size)
        prev_rank, next_rank = self.peer((self.rank - 1) % P), self.peer((self.rank + 1) % P)
        first_rank, last_rank = self.peer(0), self.peer(P - 1)
        needs_recv = lambda u: not (self.is_first and u[1] == 0)
        needs_send = lambda u: not (self.is_last and u[1] == 1)

        def p2p(*ops):
            return dist.batch_isend_irecv([dist.P2POp(fn, t, peer, group=self.group) for fn, t, peer in ops])

        recv_buf, recv_work = {}, {}        # consuming unit (item, pass) -> buffer / request group
        send_work = [None, None]
        keep_alive = [None, None]
        outs: List[Optional[torch.Tensor]] = [None] * n

        # WHEN a receive is posted.  Every rank works through the same unit sequence, rank r one tick behind rank r-1, and a
        # send completes only against the receive posted opposite it — so the receive for a message has to be in the batch
        # this rank issues at the tick the ring-predecessor SENDS it, not at the tick this rank consumes it:
        #   ranks 1 ... P-1: the predecessor (rank r-1) is at unit k+1 when this rank has done unit k, and what it sends then
        #     is this rank's own unit k+1 — send(k) and recv(k+1) in one batch, as LayerPipeline.run;
        #   rank 0: the predecessor is the LAST rank, P-1 ticks ahead on the ring — when rank 0 has done unit k, rank P-1
        #     has done unit k+1-P; if that was pass 0 of item i, its output is rank 0's input for (i, 1), consumed up to
        #     group_size - P units later (several buffers in flight when the groups are longer than the ring).
        # Posting it later (with the unit that consumes it) stops the ring as soon as groups are longer than P: rank P-1
        # blocks in a send nobody has posted a receive for, behind it every rank, and finally rank 0's own sends.
        shift = P if self.is_first else 0

        def recv_for_tick(k):
            """the receive that belongs in the batch issued after this rank's unit k (k = -1: before the first unit)"""
            kp = k + 1 - shift
            if not 0 <= kp < len(seq):
                return None
            i, ps = seq[kp]                              # the predecessor's unit of that tick
            if self.is_first:
                if ps == 1:                              # the last rank keeps pass 1's output: nothing comes
                    return None
                key = (i, 1)
            else:
                key = (i, ps)
            if key in posted:                            # (a short last group: already asked for when it was needed, below)
                return None
            posted.add(key)
            recv_buf[key] = torch.empty(shapes[i], device=device, dtype=dtype)
            return key, (dist.irecv, recv_buf[key], prev_rank)

        posted = set()

        if not feedback:
            r0 = recv_for_tick(-1)
            if r0 is not None:
                recv_work[r0[0]] = _WorkGroup(p2p(r0[1]))
        elif needs_recv(seq[0]):
            recv_buf[seq[0]] = torch.empty(shapes[seq[0][0]], device=device, dtype=dtype)
            recv_work[seq[0]] = _WorkGroup(p2p((dist.irecv, recv_buf[seq[0]], prev_rank)))
        for k, (i, ps) in enumerate(seq):
            if feedback and self.is_first and ps == 0 and i > 0:
                tok = torch.empty(shapes[i][0], 1, device=device, dtype=torch.int64)
                _WorkGroup(p2p((dist.irecv, tok, last_rank))).wait()
                token_feedback(i, tok)
            x = None
            if needs_recv(seq[k]):
                if seq[k] not in recv_work:
                    # a last group shorter than the ring: the item has not come round yet when rank 0 reaches its pass 1
                    # (a bubble the schedule cannot avoid) — ask for it now; the sender's tick has not come either
                    posted.add(seq[k])
                    recv_buf[seq[k]] = torch.empty(shapes[i], device=device, dtype=dtype)
                    recv_work[seq[k]] = _WorkGroup(p2p((dist.irecv, recv_buf[seq[k]], prev_rank)))
                recv_work.pop(seq[k]).wait()
                x = recv_buf.pop(seq[k])
            y = stage_fn(i, x, ps)
            if feedback:
                # autoregressive stream (groups of one): the next receive depends on THIS send — directly, or through the
                # token the last rank sends back — so it is issued behind the send, never in front of it
                nxt = seq[k + 1] if k + 1 < len(seq) and needs_recv(seq[k + 1]) else None
                rop = None
                if nxt is not None:
                    recv_buf[nxt] = torch.empty(shapes[nxt[0]], device=device, dtype=dtype)
                    rop = (nxt, (dist.irecv, recv_buf[nxt], prev_rank))
            else:
                rop = recv_for_tick(k)
            if not needs_send(seq[k]):
                outs[i] = y
                if feedback and i + 1 < n:
                    tok_out = token_feedback(i, y).contiguous()
                    _WorkGroup(p2p((dist.isend, tok_out, first_rank))).wait()
                if rop is not None:
                    recv_work[rop[0]] = _WorkGroup(p2p(rop[1]))
                continue
            slot = k & 1
            if send_work[slot] is not None:
                send_work[slot].wait()          # the buffer of the send two units back has left
            keep_alive[slot] = y.contiguous()
            if rop is None or feedback:
                send_work[slot] = _WorkGroup(p2p((dist.isend, keep_alive[slot], next_rank)))
                if rop is not None:
                    recv_work[rop[0]] = _WorkGroup(p2p(rop[1]))
            else:
                works = p2p((dist.isend, keep_alive[slot], next_rank), rop[1])
                if len(works) == 2:
                    send_work[slot], recv_work[rop[0]] = _WorkGroup(works[:1]), _WorkGroup(works[1:])
                else:
                    send_work[slot] = recv_work[rop[0]] = _WorkGroup(works)
        for w in send_work:
            if w is not None:
                w.wait()
        return outs


# =============================================================================
# model-level layer pipeline
# =============================================================================
class PPState:
    """Pipeline placement of a sharded HF model (``model._duo_pp`` and ``model.model._duo_pp``): which layers this
    rank kept, where they live, and the in-flight hand-off of the per-call mode."""

    def __init__(self, pipe: LayerPipeline, device, num_layers_total: int, handoff=None):
        self.pipe = pipe
        self.device = torch.device(device)
        # where hand-off buffers live: the compute device (RCCL moves device memory), or "cpu" for a gloo group —
        # the one-GPU rehearsal of the multi-rank path (every rank computes on cuda:0, the hidden state crosses
        # through host memory; tests/test_sharded_models_gpu.py, DUO_BENCH_DEBUG_SHARED_GPU)
        self.handoff = torch.device(handoff) if handoff is not None else self.device
        self.first_layer, self.last_layer = pipe.first_layer, pipe.last_layer
        self.num_layers_total = num_layers_total
        self._send = None          # (work, tensor) of the last asynchronous send of the per-call mode

    @property
    def is_first(self):
        return self.pipe.is_first

    @property
    def is_last(self):
        return self.pipe.is_last

    def local_rows(self, per_layer):
        """slice a per-layer list (head patterns, ...) given for the WHOLE model down to this stage"""
        per_layer = list(per_layer)
        if len(per_layer) == self.num_layers_total:
            return per_layer[self.first_layer:self.last_layer]
        if len(per_layer) == self.last_layer - self.first_layer:
            return per_layer
        raise ValueError(f"{len(per_layer)} per-layer rows for a stage of layers [{self.first_layer}, {self.last_layer}) "
                         f"of {self.num_layers_total}")

    # ---- per-call hand-off: ``model(input_ids=chunk, past_key_values=kv)`` on every rank ---------------------
    def recv_hidden(self, shape, dtype):
        buf = torch.empty(shape, device=self.handoff, dtype=dtype)
        dist.recv(buf, src=self.pipe.peer(self.pipe.rank - 1), group=self.pipe.group)
        return buf.to(self.device)

    def send_hidden(self, x):
        """asynchronous: the call returns while the hidden state leaves, so this rank starts its next chunk
        while the next stage works on this one (chunks pipeline across successive model calls)"""
        if self._send is not None:
            self._send[0].wait()
        x = x.contiguous().to(self.handoff)
        self._send = (dist.isend(x, dst=self.pipe.peer(self.pipe.rank + 1), group=self.pipe.group), x)

    def broadcast_from_last(self, t, shape, dtype):
        if t is None:
            t = torch.empty(shape, device=self.handoff, dtype=dtype)
        t = t.to(self.handoff)
        dist.broadcast(t, src=self.pipe.peer(self.pipe.world_size - 1), group=self.pipe.group)
        return t.to(self.device)


def shard_model_for_pp(model, device, group=None, layer_costs=None, handoff=None) -> PPState:
    """In place: keep this rank's contiguous block of decoder layers (+ ``embed_tokens`` on the first stage,
    ``norm`` / ``lm_head`` on the last), move them to ``device``, drop the rest.  Even split like the reference
    (``utils.py:251-271``) unless ``layer_costs`` is given.  Works before or after the DuoAttention enabler:
    the enablers, ``DuoAttentionStaticKVCache`` and the static model forwards all honour ``model._duo_pp``."""
    if getattr(model, "_duo_pp", None) is not None:
        return model._duo_pp
    if torch.device(device).type == "cuda":
        # one process per GPU: this rank's launches go to the CURRENT device's stream (duo_attn/_hip.py refuses tensors
        # of another device) — the reference's single-process accelerate hooks never had to choose; do it here so that
        # reference-style callers (to_device(model, devices, enable_pp=True) and nothing else) work on every rank
        torch.cuda.set_device(torch.device(device))
    inner = model.model
    n_layers = len(inner.layers)
    pipe = LayerPipeline(n_layers, group=group, layer_costs=layer_costs)
    pp = PPState(pipe, device, n_layers, handoff=handoff)
    kept = [inner.layers[i].to(pp.device) for i in range(pp.first_layer, pp.last_layer)]
    inner.layers = torch.nn.ModuleList(kept)
    if pp.is_first:
        inner.embed_tokens.to(pp.device)
    else:
        inner.embed_tokens = None
    if getattr(inner, "rotary_emb", None) is not None:
        inner.rotary_emb.to(pp.device)
    if pp.is_last:
        inner.norm.to(pp.device)
        model.lm_head.to(pp.device)
    else:
        inner.norm = None
        model.lm_head = None
    model._duo_pp = pp
    inner._duo_pp = pp
    return pp


class _StageView:
    """What DuoAttentionStaticKVCache reads from a model (config + one parameter for device / dtype)."""

    def __init__(self, model, pp):
        self.config = model.config
        self._duo_pp = pp
        self._param = next(model.model.layers[0].parameters())

    def parameters(self):
        yield self._param


class PipelinedCausalLM:
    """Explicit driver of a sharded, DuoAttention-static-patched HF Llama/Mistral (reference
    ``to_device(model, devices, enable_pp=True)``, ``duo_attn/utils.py:228-283``, as one process per GPU).

    Construct it on EVERY rank from the same model.  If the model is not sharded yet it is sharded here with the
    cost-balanced split (a layer's attention work and KV bytes grow with its retrieval heads; pass
    ``even_split_layers=True`` for the reference's even split).  ``make_kv_cache`` allocates the dual KV pools of the
    kept layers only, so a 1M-token cache is spread over the stages.

    * ``prefill(input_ids, kv, chunk, row_block=None)`` — chunked prefill, chunks streamed through the stages with
      paired send/receive groups (row blocks of a chunk when ``row_block`` is set: same mathematics, finer wavefront);
    * ``decode(first_token, kv, n_new)`` — greedy generation, the sampled token fed back from the last stage to the first;
    * ``__call__(input_ids=..., past_key_values=kv)`` — one item, like the reference's ``model(...)`` call.
    Logits exist on the last stage (``None`` elsewhere); ``decode`` returns the tokens on every rank.
    """

    def __init__(self, model, full_attention_heads, device, group=None, even_split_layers=False, handoff=None):
        heads = [[float(x) for x in torch.as_tensor(h).flatten().tolist()] for h in full_attention_heads]
        costs = None if even_split_layers else [1.0 + sum(1 for x in h if x > 0.5) / max(1, len(h)) for h in heads]
        self.pp = shard_model_for_pp(model, device, group=group, layer_costs=costs, handoff=handoff)
        self.pipe = self.pp.pipe
        self.model = model
        self.device = self.pp.device
        self.config = model.config
        self.heads = heads
        self.hidden = model.config.hidden_size
        self.dtype = next(model.model.layers[0].parameters()).dtype

    # ------------------------------------------------------------------ cache
    def make_kv_cache(self, batch_size, max_size, sink_size, recent_size):
        from .patch.static_kv_cache import DuoAttentionStaticKVCache

        return DuoAttentionStaticKVCache(_StageView(self.model, self.pp), self.heads, batch_size, max_size,
                                         sink_size, recent_size)

    # ------------------------------------------------------------------ one stage pass
    def _stage(self, x, ids, kv, pos0, row_block=None):
        inner = self.model.model
        if self.pipe.is_first:
            x = inner.embed_tokens(ids.to(self.device))
        else:
            x = x.to(self.device)          # (no-op unless the hand-off goes through host memory)
        kw = {} if row_block is None else {"row_block": row_block}
        for li, layer in enumerate(inner.layers):
            x = layer(x, position_ids=None, kv_cache=kv, layer_idx=li, use_cache=True, pos0=pos0, **kw)[0]
        return x

    def _logits(self, x):
        return self.model.lm_head(self.model.model.norm(x)[:, -1:, :])

    # ------------------------------------------------------------------ prefill
    @torch.no_grad()
    def prefill(self, input_ids, kv, chunk, row_block=None):
        """Chunked prefill of ``input_ids`` [B, N] (every rank passes the same tensor; only its shape matters
        off the first stage).  Returns the logits of the last position on the last stage, None elsewhere."""
        B, N = input_ids.shape
        items = []        # (first token, end token, (row0, chunk_len) or None)
        for c0 in range(0, N, chunk):
            c1 = min(N, c0 + chunk)
            if row_block is None or row_block >= c1 - c0:
                items.append((c0, c1, None))
            else:
                for r0 in range(0, c1 - c0, row_block):
                    items.append((c0 + r0, min(c1, c0 + r0 + row_block), (r0, c1 - c0)))
        base = kv.kv_seq_len
        shapes = [(B, b - a, self.hidden) for a, b, _ in items]
        last = {}

        def stage(i, x):
            a, b, rb = items[i]
            y = self._stage(x, input_ids[:, a:b], kv, base + a - (rb[0] if rb else 0), rb)
            if self.pipe.is_last and i == len(items) - 1:
                last["logits"] = self._logits(y)
            return y if self.pipe.is_last else y.to(self.pp.handoff)

        self.pipe.run(shapes, stage, self.pp.handoff, self.dtype)
        return last.get("logits")

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode(self, first_token, kv, n_new, return_logits=False):
        """Greedy generation of ``n_new`` tokens after ``first_token`` [B, 1] (int64, same on every rank).  Returns
        the [B, n_new] generated tokens on every rank (with ``return_logits`` also the per-step logits, which
        exist on the last stage only)."""
        B = first_token.shape[0]
        cur = {"tok": first_token.to(self.device)}
        logits_all, toks = [], []
        multi = self.pipe.world_size > 1

        def feedback(i, t):
            if self.pipe.is_last:
                return toks[-1].to(self.pp.handoff)
            cur["tok"] = t

        def stage(i, x):
            y = self._stage(x, cur["tok"], kv, kv.kv_seq_len)
            if self.pipe.is_last:
                lg = self._logits(y)
                if return_logits:
                    logits_all.append(lg)
                toks.append(lg[:, -1, :].argmax(-1, keepdim=True).to(torch.int64))
                if not multi:
                    cur["tok"] = toks[-1]
            return y if self.pipe.is_last else y.to(self.pp.handoff)

        self.pipe.run([(B, 1, self.hidden)] * n_new, stage, self.pp.handoff, self.dtype,
                      token_feedback=feedback if multi else None)
        out = torch.cat(toks, 1) if self.pipe.is_last else None
        if multi:
            out = self.pp.broadcast_from_last(out, (B, n_new), torch.int64)
        return (out, logits_all) if return_logits else out

    # ------------------------------------------------------------------ reference-style single call
    def __call__(self, input_ids=None, past_key_values=None, use_cache=True, **kw):
        """``model(input_ids=[B, S], past_key_values=kv)`` on every rank (the sharded model's own forward):
        ``.logits`` is [B, 1, V] on the last stage and None elsewhere for S > 1; for S == 1 it is broadcast."""
        return self.model(input_ids=input_ids, past_key_values=past_key_values, use_cache=use_cache, **kw)
