"""Graph-captured decode step.

The reference's decode loop (``eval/efficiency/benchmark_static.py:96-105``) re-issues ~12 launches per
layer from Python every token, with the cache lengths as Python ints baked into each one
(``static_kv_cache.py:44-45``) — at 128K the attention is long enough to hide that, at <=32K or with the
whole HF model around it the step is host-bound.  Here the lengths also live in HBM
(``DuoAttentionStaticKVCache.device_state``), the fused decode kernels read them on the device, and the
whole step — every layer's GEMMs, norms, the two attention launches per layer, the counter update and
(for the benchmark protocol) ``evict_last`` — is captured ONCE in a HIP graph and replayed per token.
The captured split-KV grid keeps working as the context grows: its balanced partition deals the *current*
number of 64-token units to the captured number of workgroups.
"""
from __future__ import annotations

from typing import Callable

import torch


class DecodeStepGraph:
    """``step_fn()`` must run ONE q_len == 1 forward through ``kv_cache`` (the patched model, or any loop
    over ``duo_static_attention_core``) using static input tensors, and may return its output tensor(s).
    Run it eagerly at least once before constructing this object (library handles, workspaces).

    ``evict_after``: tokens evicted after every step inside the graph (1 = the reference's benchmark
    protocol ``kv_cache.evict_last(1)``; 0 = real generation, the cache grows by one row per replay).
    """

    def __init__(self, kv_cache, step_fn: Callable[[], object], evict_after: int = 0):
        if kv_cache.batch_size != 1:
            raise ValueError("DecodeStepGraph supports batch size 1")
        if kv_cache.kv_seq_len < 1:
            raise ValueError("capture the decode step after the prefill (empty cache)")
        self.cache, self.step_fn, self.evict_after = kv_cache, step_fn, int(evict_after)
        kv_cache.enable_device_state()
        host = (list(kv_cache.kv_seq_len_list), list(kv_cache.streaming_kv_seq_len_list))
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.output = self._body()
        # capture records launches without running them: device state and pools are untouched, only the
        # host mirror moved while the Python code ran
        kv_cache.kv_seq_len_list[:], kv_cache.streaming_kv_seq_len_list[:] = host
        kv_cache.sync_device_state()
        self._expected = self._host_counters()

    def _host_counters(self):
        c = self.cache
        return (tuple(c.kv_seq_len_list), tuple(c.streaming_kv_seq_len_list))

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
        since the last replay without going through the graph — ``clear()``, ``evict_last()``, an eager prefill of
        the next prompt (``put_full_kv`` / ``update_streaming_kv``) — leaves that device copy stale, so the host
        counters are compared with what the last replay left and re-uploaded when they differ (one small H2D
        copy, outside the graph).  The same graph therefore serves prompt after prompt."""
        c = self.cache
        if self._host_counters() != self._expected:
            c.sync_device_state()
        if max(c.kv_seq_len_list) + 1 > c.max_size:
            raise ValueError(
                f"Trying to put 1 KVs into a cache with max size {c.max_size}, current size: {max(c.kv_seq_len_list)}."
            )
        self.graph.replay()
        W = c.sink_size + c.recent_size
        for i in range(c.num_layers):       # the host mirror follows: one step, then the eviction
            c.kv_seq_len_list[i] += 1
            c.streaming_kv_seq_len_list[i] = min(c.streaming_kv_seq_len_list[i] + 1, W)
        if self.evict_after:
            c.evict_last(self.evict_after)
        self._expected = self._host_counters()
        return self.output
