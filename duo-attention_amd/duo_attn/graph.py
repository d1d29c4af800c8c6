"""Graph-captured decode step.

The reference's decode loop (``eval/efficiency/benchmark_static.py:96-105``) re-issues ~12 launches per
layer from Python every token, with the cache lengths as Python ints baked into each one
(``static_kv_cache.py:44-45``) — at 128K the attention is long enough to hide that, at <=32K or with the
whole HF model around it the step is host-bound.  Here the lengths also live in HBM
(``DuoAttentionStaticKVCache.device_state``), the fused decode kernels read them on the device, and the
whole step — every layer's token-row linears, the two attention launches per layer, the counter update and
(for the benchmark protocol) ``evict_last`` — is captured ONCE in a HIP graph and replayed per token.
The captured split-KV grid keeps working as the context grows: its balanced partition deals the *current*
number of 64-token units to the captured number of workgroups.  The library sizes that grid from the length's BUCKET
(``duo_decode_plan_bucket``: 64-token units rounded up to a power of two), for eager and captured launches alike, so a
replay equals the eager step bit for bit anywhere inside the bucket — and when the context leaves the captured bucket
(a long generation from a short prompt; ``clear()`` and a much longer or shorter prompt through the same cache) the step is
captured again for the new one (``DecodeStepGraph.replay``): a graph captured at 1K tokens never scans a 128K pool with
the 1K grid.

Two ways in:

* ``DecodeStepGraph(kv_cache, step_fn, evict_after)`` — explicit (``tools/benchmark_static.py --graph``, tests);
* ``auto_decode_step`` — OPT-IN (``DUO_AUTO_DECODE_GRAPH=1``, or ``duo_attn.graph.AUTO_DECODE_GRAPH = True``): the patched
  ``*ForCausalLM.forward`` captures the step by itself when the call is the reference loop's decode call (one token,
  implicit positions, a static cache on this GPU, no gradients): the reference's harness runs UNCHANGED and its steps are
  graph replays after two eager ones.  OFF by default since round 6: the eager step is GPU-bound on this class of host
  (4.42 eager vs 4.44 ms/token replayed at 128K, 3.30 vs 3.33 at 32K), so the capture buys nothing there, and one abort
  inside a round-5 test run of this path was never explained ("graph-captured decode"; the soak record is
  ``profiles/r6_graph_soak.md``).  Turn it on where the host is the bottleneck: short contexts (<= 8K) or a slow /
  shared CPU.

A captured graph is never destroyed while its last replay may still be executing and never from inside somebody else's
stream capture: a retired graph goes to a module-level list with the event recorded after its last replay and is released
by a later call once that event has completed (``_retire`` / ``_drain_retired``) — no device-wide synchronisation anywhere.

Which counters the device copy currently equals is tracked on the cache (``kv_cache._device_counters``): anything that
moves the host counters without going through a graph — ``clear()``, an eager prefill, an eager decode step — makes the
next replay re-upload them (one small H2D copy outside the graph); ``evict_last`` between two replays is mirrored with
one tiny launch instead, so the reference's step-then-evict loop never synchronises the stream.
"""
from __future__ import annotations

import os
import warnings
from typing import Callable

import torch

AUTO_DECODE_GRAPH = os.environ.get("DUO_AUTO_DECODE_GRAPH", "0") == "1"
_AUTO_WARM_STEPS = 2        # eligible eager decode steps before the capture (kernels, GEMM handles and workspaces exist by then)


def _host_counters(c):
    return (tuple(c.kv_seq_len_list), tuple(c.streaming_kv_seq_len_list))


def plan_key(c):
    """what the library sizes a decode step's split-KV grids from: the bucket of the rows a retrieval head / a streaming head
    sees in the next step (cached rows + the new one) — include/duo_attn_hip.h, duo_decode_plan_bucket"""
    from . import _hip

    bucket = _hip.load_library().duo_decode_plan_bucket
    return (bucket(max(c.kv_seq_len_list) + 1), bucket(max(c.streaming_kv_seq_len_list) + 1))


class RecaptureError(RuntimeError):
    """the step could not be captured again for a new length bucket (``DecodeStepGraph.replay``); no launch was issued"""


_retired = []       # (graph object, event after its last replay): released once the event has completed


def _drain_retired(block: bool = False) -> None:
    """release retired graphs whose last replay is over.  Never while this thread is capturing (event queries are not
    capture-safe, and destroying a graph there would invalidate the capture)."""
    if not _retired or torch.cuda.is_current_stream_capturing():
        return
    keep = []
    for g, ev in _retired:
        if ev is None:
            continue        # never replayed: nothing can be executing
        if block:
            ev.synchronize()
        elif not ev.query():
            keep.append((g, ev))
    _retired[:] = keep


def _retire(graph, event) -> None:
    if graph is not None:
        _retired.append((graph, event))


class DecodeStepGraph:
    """``step_fn()`` must run ONE q_len == 1 forward through ``kv_cache`` (the patched model, or any loop
    over ``duo_static_attention_core``) using static input tensors, and may return its output tensor(s).
    Run it eagerly at least once before constructing this object (library handles, workspaces).

    ``evict_after``: tokens evicted after every step inside the graph (1 = the reference's benchmark
    protocol ``kv_cache.evict_last(1)``; 0 = real generation, the cache grows by one row per replay).
    Batch rows: every row of the cache shares the layer's counters (reference static_kv_cache.py:44-45), so a batched
    step is captured the same way (``duo_decode_layer_batched_dev_bf16``).

    Not supported: replaying two captured MODEL steps concurrently on two streams — each graph owns its split-KV scratch
    (``_hip.scratch_scope``), but torch's GEMM workspace is per capture stream and shared.  Replay them one after the other.
    """

    def __init__(self, kv_cache, step_fn: Callable[[], object], evict_after: int = 0):
        from . import _hip

        if kv_cache.kv_seq_len < 1:
            raise ValueError("capture the decode step after the prefill (empty cache)")
        self.cache, self.step_fn, self.evict_after = kv_cache, step_fn, int(evict_after)
        # this graph's own split-KV partials / tickets (_hip.scratch_scope): allocated and zeroed HERE, outside the capture,
        # so that no allocation or memset is baked into the graph (a baked memset would clear the one-launch step's sticky
        # give-up flag on every replay, and the memory would be uninitialised until the first one)
        self._scratch = {}
        _hip.prepare_graph_scratch(self._scratch, kv_cache)
        self.captures = 0
        self.graph = self.plan_key = self.output = self._last = None
        self._capture()

    def _capture(self):
        from . import _hip

        kv_cache = self.cache
        _drain_retired()
        kv_cache.enable_device_state()
        host = (list(kv_cache.kv_seq_len_list), list(kv_cache.streaming_kv_seq_len_list))
        new_key = plan_key(kv_cache)
        new_graph = torch.cuda.CUDAGraph()
        ok = False
        try:
            # (torch.cuda.graph() synchronises the device on entry: the previous capture's replays are over by the time
            # the new launches are recorded — the old object itself is only retired below, after the capture succeeded)
            with _hip.scratch_scope(self._scratch), torch.cuda.graph(new_graph):
                out = self._body()
            ok = True
        finally:
            # capture records launches without running them: device state and pools are untouched, only the
            # host mirror moved while the Python code ran
            kv_cache.kv_seq_len_list[:], kv_cache.streaming_kv_seq_len_list[:] = host
            kv_cache.use_device_state = False
            if not ok:
                # keep the previous graph and output, but make the next replay() try the capture again instead of
                # replaying a grid sized for another bucket
                self.plan_key = None
                kv_cache.sync_device_state()
        _retire(self.graph, self._last)
        self.graph, self.plan_key, self.output, self._last = new_graph, new_key, out, None
        self.captures += 1
        kv_cache.sync_device_state()

    def __del__(self):
        # the graph object outlives this one until its last replay is over (no device-wide wait, nothing while a capture
        # is in progress on this thread: this destructor can run from the cyclic GC at any point)
        try:
            _retire(self.__dict__.pop("graph", None), self.__dict__.pop("_last", None))
            _drain_retired()
        except Exception:       # (interpreter shutdown)
            pass

    def _body(self):
        c = self.cache
        c.use_device_state = True
        try:
            out = self.step_fn()
            c.device_state_add(1, 1, 1)
            if self.evict_after:
                c.device_state_add(-self.evict_after, -self.evict_after, -self.evict_after)
        finally:
            c.use_device_state = False
        return out

    def replay(self):
        """one decode step; returns the (static) output of ``step_fn`` captured at construction.

        The kernels of the captured step read the cache lengths from HBM.  Anything that moved the HOST counters
        since they were last known to agree without going through a graph — ``clear()``, an eager prefill of
        the next prompt (``put_full_kv`` / ``update_streaming_kv``) — leaves that device copy stale, so the host
        counters are compared with ``kv_cache._device_counters`` and re-uploaded when they differ (one small H2D
        copy, outside the graph).  The same graph therefore serves prompt after prompt."""
        c = self.cache
        if max(c.kv_seq_len_list) + 1 > c.max_size:
            raise ValueError(
                f"Trying to put 1 KVs into a cache with max size {c.max_size}, current size: {max(c.kv_seq_len_list)}."
            )
        if plan_key(c) != self.plan_key:
            # the context left the length bucket the split-KV grid was sized for: capture the step again for this one
            # (the eager step would plan exactly this grid, so replays stay bit-equal to it); ``self.output`` is the new
            # capture's tensor from here on
            try:
                self._capture()
            except Exception as e:      # nothing was launched and the host counters are restored: the caller may step eagerly
                raise RecaptureError(f"{type(e).__name__}: {e}") from e
        elif _host_counters(c) != c._device_counters:
            c.sync_device_state()
        self.graph.replay()
        if self._last is None:
            self._last = torch.cuda.Event()
        self._last.record()
        if _retired:
            _drain_retired()
        W = c.sink_size + c.recent_size
        for i in range(c.num_layers):       # the host mirror follows: one step, then the eviction
            c.kv_seq_len_list[i] += 1
            c.streaming_kv_seq_len_list[i] = min(c.streaming_kv_seq_len_list[i] + 1, W)
        if self.evict_after:
            hook, c._decode_graph = c._decode_graph, None       # (the graph rewound the device copy itself)
            try:
                c.evict_last(self.evict_after)
            finally:
                c._decode_graph = hook
        c._device_counters = _host_counters(c)
        return self.output

    def host_evicted(self, before, n) -> None:
        """``kv_cache.evict_last(n)`` ran on the host between two replays (the reference's benchmark loop does after
        every step, benchmark_static.py:104).  When the device counters agreed with the host's before it, one tiny launch
        rewinds them too — the next replay then needs no host-to-device copy, which would synchronise the stream once
        per token."""
        c = self.cache
        if c._device_counters is not None and before == c._device_counters:
            c.device_state_add(-n, -n, -n)
            c._device_counters = _host_counters(c)


# ------------------------------------------------------------------------------------------------------------------
# automatic capture for the reference's UNCHANGED decode loop
# ------------------------------------------------------------------------------------------------------------------
def _walk(model):
    """ONE pass over the module tree as it is NOW (plain ``_modules`` / ``_parameters`` / hook dicts — no
    ``nn.Module.__getattr__``, ~0.2 ms for a 32-layer model): (whether any module carries a forward (pre-)hook, the storage
    address of every parameter and buffer).  Walked per decode call rather than cached, so a replaced sub-module or a re-pointed
    parameter cannot hide behind a stale list."""
    import torch.nn.modules.module as nnm

    hooked = bool(nnm._global_forward_hooks or nnm._global_forward_pre_hooks)
    ptrs, stack = [], [model]
    while stack:
        m = stack.pop()
        if m._forward_hooks or m._forward_pre_hooks:
            hooked = True
        for p in m._parameters.values():
            if p is not None:
                ptrs.append(p.data_ptr())
        for b in m._buffers.values():       # (rotary inv_freq, the registered head pattern: baked into the launches as well)
            if b is not None:
                ptrs.append(b.data_ptr())
        if m._modules:
            stack.extend(c for c in m._modules.values() if c is not None)
    return hooked, tuple(ptrs)


def _model_signature(model, param_ptrs=None):
    """what the captured launches depend on besides the cache: the layers' forwards, the storage of EVERY parameter the step
    reads (q/k/v/o, gate/up/down, the norms, embed_tokens, lm_head — a captured launch bakes the raw pointers in, so a
    partial ``.data`` swap or a replaced module must retire it too), and the switches that select kernels — a change of any
    of them retires the captured step"""
    from . import _hip
    from .patch import _duo

    sig = [id(model), _duo._FUSED_DECODE_LAYER, int(_hip.load_library().duo_get_debug_flags())]
    for layer in model.model.layers:
        sig.append((id(getattr(layer.forward, "__func__", None)), id(getattr(layer.self_attn.forward, "__func__", None))))
    sig.append(_walk(model)[1] if param_ptrs is None else param_ptrs)
    return tuple(sig)


def auto_decode_eligible(model, input_ids, position_ids, past_key_values, inputs_embeds, labels, kwargs) -> bool:
    """The reference's decode call ``model(input_ids=pred, past_key_values=kv_cache, use_cache=True)``
    (eval/efficiency/benchmark_static.py:98-102): one token, one batch row, implicit positions, a non-empty static cache on
    this GPU, no gradients, a single-process model on the HIP backend."""
    from .backend import HipBackend, get_backend

    if not AUTO_DECODE_GRAPH or input_ids is None or inputs_embeds is not None or labels is not None or position_ids is not None:
        return False
    if kwargs or input_ids.shape != (1, 1) or not input_ids.is_cuda or torch.is_grad_enabled() or model.training:
        return False
    kv = past_key_values
    if kv is None or not hasattr(kv, "enable_device_state") or getattr(kv, "batch_size", 0) != 1 or kv.kv_seq_len < 1:
        return False
    if getattr(model, "_duo_pp", None) is not None or getattr(model, "_duo_tp", None) is not None:
        return False
    if getattr(kv, "_auto_graph_failed", False) or kv.device.type != "cuda" or input_ids.device != kv.device:
        return False
    if kv.device.index is not None and kv.device.index != torch.cuda.current_device():
        return False
    if torch.cuda.is_current_stream_capturing():
        return False        # somebody else's capture (DecodeStepGraph, a user's graph): this step is part of theirs
    if type(get_backend()) is not HipBackend:
        return False        # (a wrapped / recording backend observes calls from Python: never captured)
    hooked, ptrs = _walk(model)
    if hooked:
        return False        # a replay does not re-enter Python: hooks must keep firing every step
    model.__dict__["_duo_param_ptrs"] = ptrs        # (handed to auto_decode_step's signature: one walk per call)
    return True


def auto_decode_step(model, eager_forward, input_ids, kv):
    """One decode step of the unchanged reference loop through a HIP graph captured on the way: the first eligible steps
    after a change run eagerly, then the step is captured once per (model, cache, length bucket) and replayed — the cache lengths live in
    HBM (``kv.device_state``), ``kv.evict_last`` rewinds them with a launch, an eager prefill in between re-uploads them.
    Returns the logits (a fresh tensor per call), or None when this call should run eagerly."""
    st = getattr(kv, "_auto_graph", None)
    sig = _model_signature(model, model.__dict__.pop("_duo_param_ptrs", None))
    if st is None or st["sig"] != sig:
        # (a retired step's graph is released by DecodeStepGraph.__del__ -> _retire once its last replay is over)
        st = {"sig": sig, "calls": 0, "graph": None, "tok": None}
        kv._auto_graph = st
        kv._decode_graph = None
    if st["graph"] is None:
        st["calls"] += 1
        if st["calls"] <= _AUTO_WARM_STEPS:
            return None
        tok = torch.zeros(1, 1, dtype=torch.long, device=kv.device)
        tok.copy_(input_ids)
        try:
            g = DecodeStepGraph(kv, lambda: eager_forward(tok).logits, evict_after=0)
        except Exception as e:      # a layer that cannot be captured (host read-backs, foreign modules): stay eager, say so once
            kv._auto_graph_failed = True
            warnings.warn(f"DuoAttention: automatic decode-step capture failed ({type(e).__name__}: {e}); decoding eagerly. "
                          "DUO_AUTO_DECODE_GRAPH=0 disables the attempt.")
            return None
        st["graph"], st["tok"] = g, tok
        kv._decode_graph = g
    else:
        st["tok"].copy_(input_ids)
    try:
        return st["graph"].replay().clone()
    except RecaptureError as e:     # (the first capture worked, the one for the new length bucket did not: eager from here on)
        kv._auto_graph_failed, kv._decode_graph, st["graph"] = True, None, None
        warnings.warn(f"DuoAttention: re-capturing the decode step for a new context length failed ({e}); decoding eagerly.")
        return None
