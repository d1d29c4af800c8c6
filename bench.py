#!/usr/bin/env python3
"""bench.py — DuoAttention split-head attention hot path on MI355X.

Workload (BASELINE.json configs[1]): Llama-3-8B shape (32 layers, 32 q heads, 8 kv heads, D=128),
50 % streaming kv heads (per-layer retrieval-head counts of the shipped
Llama-3-8B-Instruct-Gradient-1048k pattern, seed 42, sparsity 0.5), sink 128 + recent 256,
bf16, B=1:  chunked prefill of a 131072-token context, then 128 single-token decode steps at that
context length (the reference's protocol, eval/efficiency/benchmark_static.py:68-105: decode steps
are followed by kv_cache.evict_last(1), so every step runs at the full context).

One "step" = one whole job = for every layer: RoPE -> KV append -> split-head attention ->
streaming-pool update, for every prefill chunk and every decode token.  Inputs are synthetic
N(0,1) bf16 q/k/v already resident in HBM (the q/k/v projections and MLP are hipBLASLt GEMMs
outside the path).  `value` = (prefill tokens + decode tokens) / job time; prefill tok/s and
decode tok/s are reported next to it, as is the same job with every head a retrieval head
("full attention").

Multi-GPU (--gpus N under torch.distributed.run): the 32 layers are sharded contiguously over
N ranks (strong scaling), prefill chunks are pipelined through the stages with RCCL send/recv of
the [1, C, 4096] hidden state; decode is sequential through the stages.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# retrieval ("full") kv heads per layer: attn_patterns/Llama-3-8B-Instruct-Gradient-1048k,
# seed_everything(42) + sparsify_attention_heads(sparsity=0.5)  (tests/golden/make_golden.py)
LLAMA3_8B_FULL_KV_HEADS = [1, 1, 2, 2, 2, 4, 2, 4, 6, 4, 5, 3, 2, 6, 5, 5, 5, 6, 3, 5, 6, 3, 3, 6, 4, 5, 3, 4, 6, 5, 8, 2]
# BASELINE configs[2] (Mistral-7B-Instruct-v0.2, "shipped attn_pattern"): SURVEY §8d asks for both readings of it —
# the paper's 50 % sparsity (seed 42) and the raw TSV thresholded at 0.5 (244 of 256 heads stay retrieval heads)
MISTRAL_7B_V02_FULL_KV_HEADS_50 = [3, 4, 6, 2, 4, 1, 3, 3, 2, 4, 2, 5, 6, 1, 4, 4, 5, 2, 7, 5, 6, 4, 3, 3, 4, 4, 5, 3, 4, 7, 5, 7]
MISTRAL_7B_V02_FULL_KV_HEADS_RAW = [5, 7, 8, 5, 6, 6, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 7, 8, 8, 8, 8, 8, 8, 8, 8]
PATTERNS = {
    "llama3-8b-1048k@0.5": (LLAMA3_8B_FULL_KV_HEADS, "Llama-3-8B-Instruct-Gradient-1048k shape, 50% streaming kv heads"),
    "mistral-7b-v0.2@0.5": (MISTRAL_7B_V02_FULL_KV_HEADS_50, "Mistral-7B-Instruct-v0.2 shape, shipped pattern at sparsity 0.5"),
    "mistral-7b-v0.2@raw": (MISTRAL_7B_V02_FULL_KV_HEADS_RAW, "Mistral-7B-Instruct-v0.2 shape, shipped pattern thresholded at 0.5 (95% retrieval heads)"),
}
HQ, HKV, D, HIDDEN = 32, 8, 128, 4096
SINK, RECENT = 128, 256
ROPE_THETA, ROPE_SCALE = 3580165449.0, 1.0
HBM_PEAK = 8.0e12       # B/s, MI355X_MICROARCH.md
MFMA_BF16_PEAK = 2.5e15  # FLOP/s dense


class _ShapeModel:
    def __init__(self, n_layers, device):
        import types

        self.config = types.SimpleNamespace(num_hidden_layers=n_layers, num_attention_heads=HQ,
                                            num_key_value_heads=HKV, hidden_size=HQ * D)
        self._p = torch.zeros(1, device=device, dtype=torch.bfloat16)

    def parameters(self):
        yield self._p


def decode_bytes(counts, N, W=SINK + RECENT):
    """algorithmic K+V bytes one decode token reads, per layer."""
    return [(nf * (N + 1) + (HKV - nf) * min(N + 1, W + 1)) * 2 * D * 2 for nf in counts]


def prefill_flops(counts, N, C, W=SINK + RECENT):
    """algorithmic attention FLOPs of a chunked prefill, per (chunk, layer)."""
    G = HQ // HKV
    out = []
    for s in range(0, N, C):
        c = min(C, N - s)
        row = []
        for nf in counts:
            tri = c * (c + 1) / 2
            f_full = 4 * D * (c * s + tri)
            f_str = 4 * D * (c * (min(s, W) if s > 0 else 0) + tri)
            row.append(G * (nf * f_full + (HKV - nf) * f_str) if s > 0 else G * HKV * f_full)
        out.append(row)
    return out


class HotPath:
    """The per-rank state of one job configuration (duo or full attention)."""

    def __init__(self, counts, layer_range, ctx, chunk, device):
        from duo_attn.patch.static_kv_cache import DuoAttentionStaticKVCache

        # layer_range: (first, last) — one contiguous block — or a list of blocks [(first, last), ...]: the layers of
        # the rank's virtual stages (--virtual-stages 2), run one block per pass
        blocks = [tuple(layer_range)] if isinstance(layer_range[0], int) else [tuple(b) for b in layer_range]
        self.l0, self.l1 = blocks[0][0], blocks[-1][1]
        self.counts = [c for lo, hi in blocks for c in counts[lo:hi]]
        self.pass_layers, n0 = [], 0
        for lo, hi in blocks:
            self.pass_layers.append(range(n0, n0 + hi - lo))
            n0 += hi - lo
        self.ctx, self.chunk, self.device = ctx, chunk, device
        heads = [[1.0] * nf + [0.0] * (HKV - nf) for nf in self.counts]
        self.cache = DuoAttentionStaticKVCache(_ShapeModel(len(self.counts), device), heads, 1, ctx + 5,
                                               SINK, RECENT)
        g = torch.Generator(device=device).manual_seed(1234)
        mk = lambda s, h: torch.randn(1, s, h, D, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        self.q_c, self.k_c, self.v_c = mk(chunk, HQ), mk(chunk, HKV), mk(chunk, HKV)
        self.q_1, self.k_1, self.v_1 = mk(1, HQ), mk(1, HKV), mk(1, HKV)
        self.hidden_c = torch.zeros(1, chunk, HIDDEN, device=device, dtype=torch.bfloat16)
        self.hidden_1 = torch.zeros(1, 1, HIDDEN, device=device, dtype=torch.bfloat16)
        self.chunks = [(s, min(chunk, ctx - s)) for s in range(0, ctx, chunk)]

    def free(self):
        self.cache = None
        torch.cuda.empty_cache()

    def layer_core(self, li, S, pos0, q, k, v):
        from duo_attn.patch._duo import duo_static_attention_core

        return duo_static_attention_core(q[:, :S], k[:, :S], v[:, :S], self.cache, li, pos0, ROPE_SCALE, ROPE_THETA)

    # ---- N > 1: a chunk goes through the stages in ROW BLOCKS (pipeline wavefront) -------------------
    # Row r of a chunk needs only rows <= r of the same chunk, so stage s+1 can start on the first block of a
    # chunk while stage s is still working on its later blocks: the pipeline fills in block-sized steps
    # (makespan ~ n*m + P - 1 block slots for n chunks of m blocks) instead of chunk-sized ones.  The chunk's
    # semantics are unchanged (duo_static_attention_row_block).
    def set_row_blocks(self, rows):
        self.block_rows = rows
        self.blocks = [(ci, r0, min(rows, c - r0)) for ci, (_, c) in enumerate(self.chunks) for r0 in range(0, c, rows)]

    def _layers(self, ps):
        return range(len(self.counts)) if ps is None else self.pass_layers[ps]

    # (--block-streams 2, one GPU, measurement: consecutive row blocks run on alternating HIP streams.  Block i + 1 at layer l
    #  needs block i's rows of layer l in the pools — nothing of block i's LATER layers — so (block i, layer l + 1) and
    #  (block i + 1, layer l) are independent in a real model too; issued on two streams the second launch fills the CUs the
    #  first one's tail leaves idle.  One event per (stream, layer) orders block i + 1 behind block i layer by layer.)
    def set_block_streams(self, n, serial=False):
        # (serial: the same code path and data flow on ONE side stream — what the tests compare the overlapped run with)
        one = torch.cuda.Stream(self.device) if serial else None
        self.block_streams = [one or torch.cuda.Stream(self.device) for _ in range(n)] if n > 1 else None
        self.block_events = [[torch.cuda.Event() for _ in self.counts] for _ in range(n)] if n > 1 else None

    def prefill_block_stage(self, i, x, ps=None):
        from duo_attn.patch._duo import duo_static_attention_row_block

        ci, r0, n = self.blocks[i]
        c = self.chunks[ci][1]
        streams = getattr(self, "block_streams", None)
        keep = getattr(self, "keep_outputs", None)       # (tests: a list that collects every launch's output)
        if streams and ps is None:
            k = i % len(streams)
            s, mine, prev = streams[k], self.block_events[k], self.block_events[(i - 1) % len(streams)]
            if i == 0:
                s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for j, li in enumerate(self._layers(ps)):
                    if i > 0:
                        s.wait_event(prev[j])
                    o = duo_static_attention_row_block(self.q_c[:, r0:r0 + n], self.k_c[:, r0:r0 + n], self.v_c[:, r0:r0 + n],
                                                       self.cache, li, r0, c, ROPE_SCALE, ROPE_THETA)
                    mine[j].record(s)
                    if keep is not None:
                        keep.append(o)
            return x if x is not None else self.hidden_c[:, :n]
        for li in self._layers(ps):
            o = duo_static_attention_row_block(self.q_c[:, r0:r0 + n], self.k_c[:, r0:r0 + n], self.v_c[:, r0:r0 + n],
                                               self.cache, li, r0, c, ROPE_SCALE, ROPE_THETA)
            if keep is not None:
                keep.append(o)
        return x if x is not None else self.hidden_c[:, :n]

    def prefill_stage(self, i, x, ps=None):
        s, c = self.chunks[i]
        streams = getattr(self, "block_streams", None)
        if streams and ps is None:
            # (--block-streams with whole chunks: chunk i + 1 at layer l needs chunk i's rows of layer l in the pools and nothing
            #  of chunk i's later layers — the same independence as between row blocks.  Each chunk rotates its OWN copy of the
            #  synthetic q / k rows: the whole-chunk path shares one buffer between consecutive chunks.)
            k = i % len(streams)
            st, mine, prev = streams[k], self.block_events[k], self.block_events[(i - 1) % len(streams)]
            if not hasattr(self, "qk_copies"):
                self.qk_copies = [(self.q_c.clone(), self.k_c.clone()) for _ in streams]
            q_c, k_c = self.qk_copies[k]
            if i == 0:
                for t in streams:
                    t.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for j, li in enumerate(self._layers(ps)):
                    if i > 0:
                        st.wait_event(prev[j])
                    o = self.layer_core(li, c, s, q_c, k_c, self.v_c)
                    mine[j].record(st)
                    if getattr(self, "keep_outputs", None) is not None:
                        self.keep_outputs.append(o)
            return x if x is not None else self.hidden_c[:, :c]
        for li in self._layers(ps):
            self.layer_core(li, c, s, self.q_c, self.k_c, self.v_c)
        return x if x is not None else self.hidden_c[:, :c]

    def decode_stage(self, i, x, ps=None):
        for li in self._layers(ps):
            self.layer_core(li, 1, self.ctx, self.q_1, self.k_1, self.v_1)
        if ps is None or ps == len(self.pass_layers) - 1:
            self.cache.evict_last(1)   # reference benchmark_static.py:104 (once per token: behind the rank's last block)
        return x if x is not None else self.hidden_1


def sync_all(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def run_job(hp: HotPath, pipe, n_decode, world, device, handoff=None):
    """One whole job; returns (t_total, t_prefill, t_decode) seconds (wall, max over ranks not yet taken)."""
    handoff = handoff or device
    blocked = getattr(hp, "blocks", None)
    pre_fn = hp.prefill_block_stage if blocked else hp.prefill_stage
    # (one block per rank: stage_fn(i, x); two — InterleavedLayerPipeline — stage_fn(i, x, pass): *a carries the pass)
    pre = pre_fn if handoff == device else (lambda i, x, *a: pre_fn(i, x, *a).to(handoff))
    dec = hp.decode_stage if handoff == device else (lambda i, x, *a: hp.decode_stage(i, x, *a).to(handoff))
    hp.cache.clear()
    sync_all(world)
    t0 = time.perf_counter()
    shapes = [(1, n, HIDDEN) for _, _, n in hp.blocks] if blocked else [(1, c, HIDDEN) for _, c in hp.chunks]
    pipe.run(shapes, pre, handoff)
    torch.cuda.synchronize()
    t1_local = time.perf_counter()        # this rank's own last item done (before waiting for the others)
    sync_all(world)
    t1 = time.perf_counter()
    # batch-1 decode is autoregressive: token i+1 enters stage 0 only after token i left the last stage
    # (one [1, 1] int64 hop back per token), so the layer pipeline cannot overlap decode steps
    tok = torch.zeros(1, 1, dtype=torch.int64, device=handoff)
    pipe.run([(1, 1, HIDDEN)] * n_decode, dec, handoff, token_feedback=lambda i, t: tok)
    torch.cuda.synchronize()
    t2_local = time.perf_counter()
    sync_all(world)
    t2 = time.perf_counter()
    run_job.local = (t1_local - t0, t2_local - t1)
    return t2 - t0, t1 - t0, t2 - t1


def kernel_rooflines(hp: HotPath, counts_local, n_rep=3):
    """HIP-event timing of the two attention kernels alone, on the stream they are launched on.

    decode: the split-KV kernel of every local layer at context N (merge launch suppressed with
    debug flag bit 1 so the events bracket exactly one kernel); prefill: the MFMA kernel of every
    (chunk, layer) launch of one prefill pass.  Returns dicts with summed algorithmic work and time."""
    from duo_attn import _hip
    from duo_attn.backend import get_backend

    be = get_backend()
    cache = hp.cache
    N = hp.ctx
    G = HQ // HKV
    scale = D ** -0.5
    # ---- prefill pass with per-launch events --------------------------------------------------
    cache.clear()
    pf = prefill_flops(counts_local, hp.ctx, hp.chunk)
    evs = []
    for ci, (s, c) in enumerate(hp.chunks):
        for li, nf in enumerate(counts_local):
            q, k, v = hp.q_c[:, :c], hp.k_c[:, :c], hp.v_c[:, :c]
            fk, fv, sk, sv = cache.split_kv(li, k, v)
            past_l = cache.kv_seq_len_list[li]
            cache.put_full_kv(li, fk, fv)
            out = torch.empty_like(q)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if s == 0:
                e0.record()
                be.attention(q[0], out[0], G, (HKV, 0, None, (k[0], v[0])), None, scale)
                e1.record()
            else:
                ns = HKV - nf
                pk, pv = cache.full_key_states_list[li], cache.full_value_states_list[li]
                ck, cv = cache.get_streaming_kv(li)
                full = (nf, 0, (pk[0, :past_l], pv[0, :past_l]), (pk[0, past_l:past_l + c], pv[0, past_l:past_l + c])) if nf else None
                stream = (ns, nf * G, (ck[0], cv[0]), (sk[0], sv[0])) if ns else None
                e0.record()
                be.attention(q[0], out[0], G, full, stream, scale)
                e1.record()
            cache.update_streaming_kv(li, sk, sv)
            evs.append((e0, e1))
    torch.cuda.synchronize()
    t_prefill = sum(a.elapsed_time(b) for a, b in evs) * 1e-3
    prefill = {"flops": float(sum(sum(r) for r in pf)), "seconds": t_prefill, "launches": len(evs)}

    # ---- decode split kernel at context N ------------------------------------------------------
    db = decode_bytes(counts_local, N)
    evs = []
    _hip.set_debug_flags(2)   # bit 1: no merge launch -> events bracket the split kernel only
    try:
        q = hp.q_1
        out = torch.empty_like(q)
        for rep in range(n_rep + 1):
            for li, nf in enumerate(counts_local):
                ns = HKV - nf
                pk, pv = cache.full_key_states_list[li], cache.full_value_states_list[li]
                sk, sv = cache.streaming_key_states_list[li], cache.streaming_value_states_list[li]
                W = SINK + RECENT
                full = (nf, 0, (pk[0, :N], pv[0, :N]), (pk[0, N:N + 1], pv[0, N:N + 1])) if nf else None
                stream = (ns, nf * G, (sk[0, :W], sv[0, :W]), (hp.k_1[0, :, nf:], hp.v_1[0, :, nf:])) if ns else None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                be.attention(q[0], out[0], G, full, stream, scale)
                e1.record()
                if rep > 0:
                    evs.append((e0, e1))
        torch.cuda.synchronize()
    finally:
        _hip.set_debug_flags(0)
    t_dec = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / n_rep
    decode = {"bytes": float(sum(db)), "seconds": t_dec, "launches": len(counts_local)}
    return prefill, decode


def decode_graph_leg(hp: HotPath, counts_local, n=40):
    """The same decode step (every local layer through duo_static_attention_core, then evict_last(1)) captured once in a
    HIP graph and replayed: the GPU-side cost of the step with no host in the loop.  At 128K the eager loop of the timed
    region is GPU-bound and the two figures agree; at <= 64K it is host-bound (its time does not depend on the context) and
    this is what the kernels cost.  Extra information, outside the timed region."""
    from duo_attn.graph import DecodeStepGraph

    cache = hp.cache
    for li in range(len(counts_local)):     # (the timed jobs left the cache at the full context; contents do not matter here)
        cache.kv_seq_len_list[li] = hp.ctx
        cache.streaming_kv_seq_len_list[li] = min(hp.ctx, SINK + RECENT)

    def step():
        for li in range(len(counts_local)):
            hp.layer_core(li, 1, hp.ctx, hp.q_1, hp.k_1, hp.v_1)

    for _ in range(2):
        step()
        cache.evict_last(1)
    g = DecodeStepGraph(cache, step, evict_after=1)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    by = float(sum(decode_bytes(counts_local, hp.ctx)))
    return {"ms_per_token": ms, "achieved": by / (ms * 1e-3) / 1e9, "unit": "GB/s", "frac": by / (ms * 1e-3) / HBM_PEAK,
            "what": "the decode step of the timed region as ONE captured HIP graph (duo_attn.graph.DecodeStepGraph, evict_last(1) "
                    "inside): GPU-side cost, no host in the loop"}


def cpu_baseline(counts, ctx, chunk, n_decode):
    """Oracle (torch CPU restatement of the reference semantics, fp32 math) timed on this host's
    cores on a bounded sample, scaled to the job by algorithmic work."""
    from oracle.duo_oracle import flash_attn_func_ref

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    nf = 4
    # decode sample: one layer with 4 retrieval kv heads at the full context
    N = ctx
    q = torch.randn(1, 1, nf * 4, D, generator=g).to(torch.bfloat16)
    k = torch.randn(1, N + 1, nf, D, generator=g).to(torch.bfloat16)
    v = torch.randn(1, N + 1, nf, D, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    flash_attn_func_ref(q, k, v)
    t_dec = time.perf_counter() - t0
    bytes_sample = nf * (N + 1) * 2 * D * 2
    dec_tok_s = 1.0 / (t_dec * sum(decode_bytes(counts, ctx)) / bytes_sample)
    # prefill sample: 128 query rows of one retrieval kv-head group against a 32768-token past
    Sq, past = 128, min(32768, ctx)
    q = torch.randn(1, Sq, 4, D, generator=g).to(torch.bfloat16)
    k = torch.randn(1, past + Sq, 1, D, generator=g).to(torch.bfloat16)
    v = torch.randn(1, past + Sq, 1, D, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    flash_attn_func_ref(q, k, v)
    t_pre = time.perf_counter() - t0
    flops_sample = 4 * 4 * D * (Sq * past + Sq * (Sq + 1) / 2)
    total_flops = sum(sum(r) for r in prefill_flops(counts, ctx, chunk))
    pre_tok_s = ctx / (t_pre * total_flops / flops_sample)
    job_tok_s = (ctx + n_decode) / (ctx / pre_tok_s + n_decode / dec_tok_s)
    # BASELINE configs[0] at its stated size, attention path on the host: Llama-2-7B shape (MHA, 32 kv heads), one
    # 4096-token prompt (first chunk: every head causal, reference llama.py:364-372); 4 of the 32 identical
    # layers are timed and the figure is scaled by 8
    H2, N2, L2_TIMED = 32, 4096, 4
    q = torch.randn(1, N2, H2, D, generator=g).to(torch.bfloat16)
    k = torch.randn(1, N2, H2, D, generator=g).to(torch.bfloat16)
    v = torch.randn(1, N2, H2, D, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    for _ in range(L2_TIMED):
        flash_attn_func_ref(q, k, v)
    t_cfg1 = (time.perf_counter() - t0) * (32 / L2_TIMED)
    return {
        "value": job_tok_s,
        "unit": "tokens/s",
        "cores": cores,
        "kind": "port",
        "prefill_tok_s": pre_tok_s,
        "decode_tok_s": dec_tok_s,
        "cfg1_llama2_7b_4k_prefill_tok_s": N2 / t_cfg1,
        "cfg1_sample": f"Llama-2-7B shape, 4096-token single-chunk prefill, 32 q = kv heads: {L2_TIMED} of 32 layers timed "
                       f"({t_cfg1 * L2_TIMED / 32:.2f} s), x{32 // L2_TIMED}",
        "sample": (f"oracle/duo_oracle.py flash_attn_func_ref on {cores} threads: decode = 1 layer, 4 retrieval kv "
                   f"heads x {N + 1} keys ({t_dec:.2f} s); prefill = 128 rows x 4 q heads vs {past + Sq} keys "
                   f"({t_pre:.2f} s); scaled to the 32-layer job by algorithmic bytes / FLOPs"),
    }


def measure_traffic(args):
    """FETCH_SIZE of the two attention kernels, measured now: this file re-runs itself as `--traffic-probe` (one
    prefill pass + 4 decode steps) under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (a counter pass of its own, no
    other trace domain), the rocpd database is read back and the per-launch mean is corrected x2 x 1024 B (FETCH_SIZE
    counts KiB and reads half of a wide coalesced stream on gfx950 — MI355X_MICROARCH.md, HBM).  Returns
    ({kernel: bytes per launch}, provenance string)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "unavailable: rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="duo_traffic_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", out, "-o", "t", "--", sys.executable,
           os.path.abspath(__file__), "--traffic-probe", "--ctx", str(args.ctx), "--chunk", str(args.chunk),
           "--layers", str(args.layers), "--pattern", args.pattern] + (["--no-int4"] if args.no_int4 else [])
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return {}, f"unavailable: rocprofv3 child rc={r.returncode}: {(r.stderr or r.stdout)[-200:]!r}"
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select kernel_name, avg(value), count(*) from counters_collection "
                           "where counter_name = 'FETCH_SIZE' group by kernel_name").fetchall()
        # one figure per kernel family = launch-weighted mean (prefill: duo_prefill_w64_kernel, every launch of this
        # workload; duo_prefill_kernel, the 8-wave A/B kernel, would be counted with it)
        res, n, acc = {}, {}, {}
        for name, mean, cnt in rows:
            key = "duo_prefill" if (("duo_prefill_w64_kernel" in name or "duo_prefill_kernel" in name) and "f16" not in name) else (
                "duo_decode_scan_kernel" if ("duo_decode_scan_kernel" in name or "duo_decode_split_kernel" in name) else (
                    "duo_int4_decode" if ("duo_int4_decode_fold_kernel" in name or "duo_int4_decode_mfma_kernel" in name) else (
                        "duo_token_linear_kernel" if "duo_token_linear_kernel" in name else None)))
            if key:
                tot, c0 = acc.get(key, (0.0, 0))
                acc[key] = (tot + float(mean) * cnt, c0 + cnt)
        for key, (tot, c0) in acc.items():
            res[key] = tot / c0 * 1024.0 * 2.0
            n[key] = c0
        if not res:
            return {}, "unavailable: no FETCH_SIZE rows for the attention kernels in the rocpd database"
        return res, (f"rocprofv3 --pmc FETCH_SIZE --kernel-trace child of this run ({time.strftime('%Y-%m-%d %H:%M:%S')}), "
                     f"mean over {n} launches, x2 gfx950 correction")
    except Exception as e:     # a profiler problem must never cost the bench line
        return {}, f"unavailable: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def traffic_probe(args, device):
    """child of measure_traffic: one prefill pass and a few decode steps, nothing timed"""
    from duo_attn.pipeline import InterleavedLayerPipeline, LayerPipeline

    counts = PATTERNS[args.pattern][0][: args.layers]
    hp = HotPath(counts, (0, len(counts)), args.ctx, args.chunk, device)
    run_job(hp, LayerPipeline(len(counts), rank=0, world_size=1), 4, 1, device)
    torch.cuda.synchronize()
    hp.free()
    if not args.no_int4:
        int4_leg(device, reps=2, parity=False, prefill=False)     # a few launches of the INT4 decode kernel for FETCH_SIZE
    # the decode step's token-row linears: the four launches of 4 distinct layers, eager (every launch a dispatch)
    Ws, h0, _, layer, _, _ = _token_linear_setup(device, 4)
    for w in Ws:
        layer(w, h0)
    torch.cuda.synchronize()


def live_parity(device):
    """Measured error of the two attention kernels at the bench workload's own sizes, against an exact-P fp32
    attention written in plain torch on the same GPU inputs (sampled query rows; an independent computation, not the
    oracle): prefill = the last chunk of the 131072-token context on a layer with 4 retrieval + 4 streaming kv
    heads, decode = one token over 131072 cached rows.  rms(err)/rms(ref) of exact bf16 arithmetic with P rounded to
    bf16 (FA2's and the MFMA kernel's) is sqrt(2) * 1.65e-3 = 2.3e-3, with fp32 P (decode) 1.65e-3."""
    from duo_attn.backend import get_backend

    be = get_backend()
    g = torch.Generator(device=device).manual_seed(7)
    G, nf, ns, W, scale = HQ // HKV, 4, 4, SINK + RECENT, D ** -0.5
    rnd = lambda *sh: torch.randn(*sh, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    pool = lambda h, rows: rnd(h, rows, D).permute(1, 0, 2)          # head-major storage, token-major view

    def ref_rows(q_rows, K, V, vis):
        s_ = torch.einsum("ngd,td->ngt", q_rows.float(), K.float()) * scale
        t = torch.arange(K.shape[0], device=device)
        s_.masked_fill_(t[None, None, :] >= vis[:, None, None], float("-inf"))
        return torch.einsum("ngt,td->ngd", torch.softmax(s_, -1), V.float())

    def stats(o, r):
        e = (o.float() - r).abs()
        return {"rms_err_over_rms_ref": float((o.float() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()),
                "max_abs_err": float(e.max()), "rms_ref": float(r.pow(2).mean().sqrt()), "n": int(r.numel())}

    S, past = 16384, 131072 - 16384
    q, kn, vn = rnd(S, HQ, D), pool(HKV, S), pool(HKV, S)
    fk, fv, sk, sv = pool(nf, past + S), pool(nf, past + S), pool(ns, W), pool(ns, W)
    fk[past:] = kn[:, :nf]
    fv[past:] = vn[:, :nf]
    out = torch.empty_like(q)
    be.attention(q, out, G, (nf, 0, (fk[:past], fv[:past]), (fk[past:], fv[past:])),
                 (ns, nf * G, (sk, sv), (kn[:, nf:], vn[:, nf:])), scale)
    rows = torch.randint(0, S, (32,), generator=g, device=device)
    ref = torch.empty(32, HQ, D, device=device)
    for h in range(nf):
        ref[:, h * G:(h + 1) * G] = ref_rows(q[rows, h * G:(h + 1) * G], fk[:, h], fv[:, h], past + rows + 1)
    for j in range(ns):
        h = nf + j
        ref[:, h * G:(h + 1) * G] = ref_rows(q[rows, h * G:(h + 1) * G], torch.cat([sk[:, j], kn[:, h]]),
                                             torch.cat([sv[:, j], vn[:, h]]), W + rows + 1)
    res = {"prefill_chunk16384_past114688": stats(out[rows], ref)}
    # decode over the same pools: N cached rows + the new row
    N = past + S - 1
    q1 = rnd(1, HQ, D)
    o1 = torch.empty_like(q1)
    be.attention(q1, o1, G, (nf, 0, (fk[:N], fv[:N]), (fk[N:N + 1], fv[N:N + 1])),
                 (ns, nf * G, (sk[:W - 1], sv[:W - 1]), (sk[W - 1:], sv[W - 1:])), scale)
    one = torch.ones(1, dtype=torch.long, device=device)
    r1 = torch.empty(1, HQ, D, device=device)
    for h in range(nf):
        r1[:, h * G:(h + 1) * G] = ref_rows(q1[:, h * G:(h + 1) * G], fk[:N + 1, h], fv[:N + 1, h], one * (N + 1))
    for j in range(ns):
        h = nf + j
        r1[:, h * G:(h + 1) * G] = ref_rows(q1[:, h * G:(h + 1) * G], sk[:, j], sv[:, j], one * W)
    res["decode_131072"] = stats(o1, r1)
    res["reference"] = "exact-P fp32 softmax attention in torch on the GPU, 32 sampled query rows x 32 q heads (prefill), all q heads (decode)"
    res["bar"] = "tests/helpers.py attn_close: rms(err) <= 2.5e-3 rms(ref) + elementwise budget; per-config figures of the GPU test-suite: profiles/parity_r6.json"
    return res


def _exact_rows(q_rows, K, V, vis, scale):
    """exact-P fp32 softmax attention in plain torch: q_rows [n, g, D], K / V [T, D], row i sees keys [0, vis[i])"""
    s_ = torch.einsum("ngd,td->ngt", q_rows.float(), K.float()) * scale
    t = torch.arange(K.shape[0], device=K.device)
    s_.masked_fill_(t[None, None, :] >= vis[:, None, None], float("-inf"))
    return torch.einsum("ngt,td->ngd", torch.softmax(s_, -1), V.float())


def _token_linear_setup(device, n_layers):
    """weights of `n_layers` Llama-3-8B decoder layers (random, bf16) and the four token-row linear launches of one layer"""
    from duo_attn import _hip

    Hd, I, NQ, NKV = HIDDEN, 14336, HQ * D, HKV * D
    g = torch.Generator(device=device).manual_seed(5)
    rn = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g, device=device, dtype=torch.float32) * sc).to(torch.bfloat16)
    Ws = [dict(q=rn(NQ, Hd, sc=Hd ** -0.5), k=rn(NKV, Hd, sc=Hd ** -0.5), v=rn(NKV, Hd, sc=Hd ** -0.5), o=rn(Hd, NQ, sc=NQ ** -0.5),
               g=rn(I, Hd, sc=Hd ** -0.5), u=rn(I, Hd, sc=Hd ** -0.5), d=rn(Hd, I, sc=I ** -0.5), n1=rn(Hd).abs() + 0.5, n2=rn(Hd).abs() + 0.5)
          for _ in range(n_layers)]
    h0, ao = rn(1, Hd), rn(1, NQ)
    layer_bytes = (NQ + 2 * NKV + NQ + 3 * I) * Hd * 2

    def layer(w, h):
        _hip.token_linear(h, [(w["q"], None), (w["k"], None), (w["v"], None)], norm=(w["n1"], 1e-5))     # (attention op between: not timed here)
        h1 = _hip.token_linear(ao, [(w["o"], None)], residual=h)
        gu = _hip.token_linear(h1, [(w["g"], None), (w["u"], None)], norm=(w["n2"], 1e-5))
        return _hip.token_linear(gu[:, :I], [(w["d"], None)], x2=gu[:, I:], residual=h1)

    return Ws, h0, ao, layer, layer_bytes, I


def token_linear_leg(device, n_layers=32, reps=20):
    """The decode step's token-row linears (DESIGN row (g), csrc/duo_linear.hip), driver-timed: the four launches of a
    Llama-3-8B decoder layer (q|k|v with the RMSNorm prologue, o_proj + residual, gate|up with the RMSNorm prologue,
    down_proj with the SiLU*mul prologue + residual) over `n_layers` DISTINCT weight sets (436 MB each: nothing is served
    from L2 / MALL), captured once in a HIP graph and replayed — device time without host launch overhead, the way the
    model-level decode step runs them.  HBM-bound: algorithmic bytes = the weight bytes.  Live parity of one layer's
    chain against the same modules through torch (library GEMMs, separate norm / activation / add kernels)."""
    import torch.nn.functional as F
    from duo_attn import _hip

    Ws, h0, ao, layer, layer_bytes, I = _token_linear_setup(device, n_layers)

    def chain():
        h = h0
        for w in Ws:
            h = layer(w, h)
        return h

    for _ in range(2):
        chain()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        chain()
    for _ in range(3):
        graph.replay()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / len(evs)
    # live parity: one layer through torch's own modules (flashinfer-form RMSNorm = duo_rmsnorm_bf16)
    w = Ws[0]
    got = layer(w, h0)
    h1 = h0 + F.linear(ao, w["o"])
    xn2 = _hip.rmsnorm(h1, w["n2"], 1e-5)
    want = h1 + F.linear(F.silu(F.linear(xn2, w["g"])) * F.linear(xn2, w["u"]), w["d"])
    rel = ((got.float() - want.float()).norm() / want.float().norm()).item()
    gbps = layer_bytes * n_layers / t / 1e9
    return {"what": "q|k|v, o_proj, gate|up, down_proj of a Llama-3-8B decoder layer at q_len == 1 (duo_token_linear_bf16), "
                    f"{n_layers} distinct layers in one captured graph",
            "layers": n_layers, "launches_per_layer": 4, "algorithmic_bytes_per_layer": float(layer_bytes),
            "us_per_layer": t / n_layers * 1e6, "ms_per_token_32_layers": t / n_layers * 32 * 1e3,
            "roofline": {"kernel": "duo_token_linear_kernel", "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": gbps * 1e9 / HBM_PEAK, "traffic": None,
                         "avg_launch_ms": t / n_layers / 4 * 1e3, "algorithmic_bytes_per_launch": layer_bytes / 4.0},
            "parity_vs_torch_modules_rel_l2": rel}


def int4_leg(device, ctx=1048576, reps=5, parity=True, prefill=True):
    """BASELINE configs[4] (INT4 KV pools), driver-timed: (1) `duo_int4_decode_mfma_kernel` — the fused decode attention
    of ONE layer (4 retrieval + 4 streaming kv heads) straight on the packed pools at a 1 M-token context: 136 B per
    (token, kv head) instead of the reference's dequantise-everything + flash_attn_func (demo/int4_kv.py:373-436,
    demo/w8a8kv4_llama.py:240-274), 570 MB per launch, HBM-bound; (2) one fp16 chunked-prefill launch (chunk 16384 at past
    114688) over DEQUANTISED pools, which is how the reference prefills; (3) live parity of the decode kernel at 131072
    context against exact fp32 attention over the pools dequantised in plain torch fp16 arithmetic (hmul then hadd)."""
    from duo_attn import _hip

    G, nf, ns, W = HQ // HKV, 4, 4, SINK + RECENT
    g = torch.Generator(device=device).manual_seed(11)
    scale = D ** -0.5

    def pools(h, T):
        q_ = torch.randint(0, 256, (h, T, 64), generator=g, device=device, dtype=torch.uint8).permute(1, 0, 2)
        sz = (torch.rand(h, T, 2, generator=g, device=device) * 0.3 + 0.01).to(torch.float16).permute(1, 0, 2)
        return q_, sz

    res = {}
    # ---- (1) decode at `ctx`
    fkq, fksz = pools(nf, ctx + 1)
    fvq, fvsz = pools(nf, ctx + 1)
    skq, sksz = pools(ns, W + 1)
    svq, svsz = pools(ns, W + 1)
    full = _hip.make_int4_pool(fkq, fksz, fvq, fvsz, ctx + 1, 0)
    stream = _hip.make_int4_pool(skq, sksz, svq, svsz, W + 1, nf * G)
    q = torch.randn(HQ, D, generator=g, device=device).to(torch.float16)
    out = torch.empty_like(q)
    rows = nf * (ctx + 1) + ns * (W + 1)
    nbytes = rows * 2 * 68

    def timed(flags, mode):
        _hip.set_debug_flags(flags)
        try:
            evs = []
            for i in range(reps + 2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _hip.attn_decode_int4(q, out, G, full, stream, scale, fused=mode)
                b.record()
                if i >= 2:
                    evs.append((a, b))
            torch.cuda.synchronize()
        finally:
            _hip.set_debug_flags(0)
        return sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / len(evs)

    # mode 0 = the dequantising kernel (bit-exact reference values per element: the default of DuoAttentionStaticINT4KVCache);
    # mode 2 = the opt-in folded kernel (scale / zero applied to the score tile and to P, folded_decode=True)
    t_kernel = timed(2, 0)     # debug bit 1: no merge launch -> the events bracket the split kernel alone
    t_op = timed(0, 0)         # split + merge: the whole attention of the step
    t_kernel_f, t_op_f = timed(2, 2), timed(0, 2)
    res["decode"] = {"context": ctx, "algorithmic_bytes_per_launch": float(nbytes), "rows": rows,
                     "kernel": "duo_int4_decode_mfma_kernel",
                     "kernel_ms": t_kernel * 1e3, "op_ms": t_op * 1e3,
                     "kernel_GBps": nbytes / t_kernel / 1e9, "op_GBps": nbytes / t_op / 1e9,
                     "rows_per_us": rows / t_op / 1e6,
                     "bf16_equivalent_GBps": rows * 512 / t_op / 1e9,
                     "folded_opt_in": {"kernel": "duo_int4_decode_fold_kernel", "kernel_ms": t_kernel_f * 1e3,
                                       "op_ms": t_op_f * 1e3, "kernel_GBps": nbytes / t_kernel_f / 1e9,
                                       "frac": nbytes / t_kernel_f / HBM_PEAK,
                                       "note": "no per-element dequantisation in tame tiles; outputs within one fp16 ulp per "
                                               "dequantised value of the default's, outside its strict bar (DESIGN §3)"}}
    del fkq, fksz, fvq, fvsz, full
    torch.cuda.empty_cache()
    if prefill:
        # ---- (2) fp16 prefill over dequantised pools: quantise real rows, dequantise them (as the reference's get()), attend
        from duo_attn.backend import get_backend

        be = get_backend()
        S, past = 16384, 131072 - 16384
        mk = lambda h, T: torch.randn(h, T, D, generator=g, device=device, dtype=torch.float32).to(torch.float16).permute(1, 0, 2)
        kq, ksz = pools(nf, past + S)
        vq, vsz = pools(nf, past + S)
        _hip.int4_quantize(mk(nf, past + S), kq, ksz, 0)
        _hip.int4_quantize(mk(nf, past + S), vq, vsz, 0)
        fk = _hip.int4_dequantize(kq, ksz, past + S, torch.empty((past + S) * nf * D, dtype=torch.float16, device=device))
        fv = _hip.int4_dequantize(vq, vsz, past + S, torch.empty((past + S) * nf * D, dtype=torch.float16, device=device))
        sk, sv = mk(ns, W + S).contiguous(), mk(ns, W + S).contiguous()      # streaming class: pool (W rows) ++ chunk
        qc = torch.randn(S, HQ, D, generator=g, device=device).to(torch.float16)
        oc = torch.empty_like(qc)
        cls_full = (nf, 0, (fk[:past], fv[:past]), (fk[past:], fv[past:]))
        cls_str = (ns, nf * G, (sk[:W], sv[:W]), (sk[W:], sv[W:]))
        evs = []
        for i in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            be.attention(qc, oc, G, cls_full, cls_str, scale)
            b.record()
            if i >= 1:
                evs.append((a, b))
        torch.cuda.synchronize()
        t_pre = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / len(evs)
        tri = S * (S + 1) / 2
        flops = 4 * D * G * (nf * (S * past + tri) + ns * (S * W + tri))
        res["prefill_f16_over_dequantised_pools"] = {
            "kernel": "duo_prefill_w64_f16_kernel", "chunk": S, "past": past, "launch_ms": t_pre * 1e3,
            "TFLOPs": flops / t_pre / 1e12, "frac_of_2.5PF": flops / t_pre / MFMA_BF16_PEAK}
        if parity:
            # ---- (3) decode over the SAME quantised rows at 131072 context vs exact attention over torch-dequantised values
            N = past + S
            skq2, sksz2 = pools(ns, W)
            svq2, svsz2 = pools(ns, W)
            _hip.int4_quantize(mk(ns, W), skq2, sksz2, 0)
            _hip.int4_quantize(mk(ns, W), svq2, svsz2, 0)
            q1 = torch.randn(HQ, D, generator=g, device=device).to(torch.float16)
            o1 = torch.empty_like(q1)
            _hip.attn_decode_int4(q1, o1, G, _hip.make_int4_pool(kq, ksz, vq, vsz, N, 0),
                                  _hip.make_int4_pool(skq2, sksz2, svq2, svsz2, W, nf * G), scale)

            def deq(qp, szp):      # [T, h, 64] u8, [T, h, 2] f16 -> [T, h, 128] f16: hi nibble first, hmul then hadd in fp16
                codes = torch.stack([qp >> 4, qp & 15], -1).flatten(-2).to(torch.float16)
                return (codes * szp[..., 0:1]) + szp[..., 1:2]

            ref = torch.empty(1, HQ, D, device=device)
            one = torch.ones(1, dtype=torch.long, device=device)
            for h in range(nf):
                ref[:, h * G:(h + 1) * G] = _exact_rows(q1[None, h * G:(h + 1) * G], deq(kq[:, h], ksz[:, h]),
                                                        deq(vq[:, h], vsz[:, h]), one * N, scale)
            for j in range(ns):
                h = nf + j
                ref[:, h * G:(h + 1) * G] = _exact_rows(q1[None, h * G:(h + 1) * G], deq(skq2[:, j], sksz2[:, j]),
                                                        deq(svq2[:, j], svsz2[:, j]), one * W, scale)
            e = o1[None].float() - ref
            res["parity_decode_131072"] = {
                "rms_err_over_rms_ref": float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
                "max_abs_err": float(e.abs().max()), "rms_ref": float(ref.pow(2).mean().sqrt()),
                "reference": "exact-P fp32 softmax attention in torch over the pools dequantised in torch fp16 (hmul, hadd); "
                             "fp16 output rounding alone is 2^-11 / sqrt(3) = 2.8e-4 relative"}
    return res


def _int4_cache(device, counts, max_size, chunk):
    """DuoAttentionStaticINT4KVCache of the job's head pattern over pools of plausible rows: random nibbles, scales in
    (0.01, 0.31), zeros ~ N(0, 1) (timing and the scale vote of the decode kernel see ordinary rows, not zeros)"""
    from duo_attn.int4_kv import DuoAttentionStaticINT4KVCache

    class _M:
        def __init__(self):
            import types

            self.config = types.SimpleNamespace(num_hidden_layers=len(counts), num_attention_heads=HQ,
                                                num_key_value_heads=HKV, hidden_size=HQ * D)
            self._p = torch.zeros(1, device=device, dtype=torch.float16)

        def parameters(self):
            yield self._p

    heads = [[1.0] * nf + [0.0] * (HKV - nf) for nf in counts]
    cache = DuoAttentionStaticINT4KVCache(_M(), heads, 1, max_size, SINK, RECENT, chunk)
    for caches in (cache.full_key_caches, cache.full_value_caches, cache.streaming_key_caches, cache.streaming_value_caches):
        for c in caches:
            if c.quantized_data.numel():
                c.quantized_data.random_(0, 256)
                c.scale_zero[..., 0].uniform_(0.01, 0.31)
                c.scale_zero[..., 1].normal_(0.0, 1.0)
    return cache


def int4_whole_step(device, counts, ctx=3_300_000, steps=4):
    """BASELINE configs[4] as written: the decode step of ALL 32 layers (the shipped pattern at 50 %) over INT4 pools at a
    3.3 M-token context — 128 retrieval kv heads x 3.3 M rows + 128 streaming heads x 385 rows, 136 B of packed K + V per row
    = 57.4 GB read per generated token (the bf16 pools of the same context would be 216 GB) — through
    DuoAttentionStaticINT4KVCache.decode_attention (scan + merge launch per layer), HIP events around whole steps."""
    cache = _int4_cache(device, counts, ctx + 1, 1)
    W = SINK + RECENT
    for l in range(len(counts)):
        cache.kv_seq_len_list[l] = ctx + 1              # (after put(): the new token's row is in the pools)
        cache.streaming_kv_seq_len_list[l] = W + 1
    g = torch.Generator(device=device).manual_seed(3)
    q = torch.randn(1, 1, HQ, D, generator=g, device=device).to(torch.float16)
    rows = sum(nf * (ctx + 1) + (HKV - nf) * (W + 1) for nf in counts)
    nbytes = rows * 2 * 68

    def step():
        for l in range(len(counts)):
            cache.decode_attention(l, q)

    def timed_steps():
        step()
        evs = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            step()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / len(evs)

    t = timed_steps()                      # the default: every tile dequantised in registers (duo_int4_decode_mfma_kernel)
    cache.folded_decode = True
    t_f = timed_steps()                    # opt-in folded decode (duo_int4_decode_fold_kernel)
    res = {"context": ctx, "layers": len(counts), "retrieval_kv_heads": int(sum(counts)), "packed_bytes_per_token": float(nbytes),
           "pool_bytes": int(cache.memory_usage), "ms_per_token": t * 1e3, "achieved": nbytes / t / 1e9, "unit": "GB/s",
           "frac": nbytes / t / HBM_PEAK, "launches_per_token": 2 * len(counts), "rows_per_us": rows / t / 1e6,
           "bf16_equivalent_GBps": rows * 512 / t / 1e9,
           "folded_opt_in": {"ms_per_token": t_f * 1e3, "achieved": nbytes / t_f / 1e9, "frac": nbytes / t_f / HBM_PEAK},
           "what": "32-layer decode step over INT4 pools, shipped pattern, whole step incl. the merge launches"}
    del cache
    torch.cuda.empty_cache()
    return res


def int4_prefill_chunk(device, counts, ctx=131072, chunk=16384, reps=2):
    """One chunk of the INT4 chunked prefill as the reference runs it (demo/int4_kv.py:261-436 around
    demo/w8a8kv4_llama.py:226-274), all 32 layers of the pattern: put() quantises the chunk's K / V rows into the pools,
    get() dequantises the WHOLE pools into fp16 scratch, flash attention over the scratch (the fp16 MFMA prefill kernel),
    compress() slides the streaming pools.  The LAST chunk of a `ctx`-token prompt (past = ctx - chunk): tokens per second
    of the attention path at that depth, and the share of the quantise / dequantise passes."""
    cache = _int4_cache(device, counts, ctx + 8, chunk)
    past, W = ctx - chunk, SINK + RECENT
    g = torch.Generator(device=device).manual_seed(4)
    mk = lambda h: torch.randn(1, chunk, h, D, generator=g, device=device, dtype=torch.float32).to(torch.float16)
    q, k, v = mk(HQ), mk(HKV), mk(HKV)

    def one_chunk(attend=True):
        for l in range(len(counts)):
            cache.kv_seq_len_list[l] = past
            cache.streaming_kv_seq_len_list[l] = W
            cache.put(l, k, v, dequantize=False)
            if attend:
                cache.prefill_attention(l, q, k, v)
            else:
                cache.get(l)
            cache.compress(l)

    def timed(fn):
        fn()
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / len(evs)

    t_all = timed(one_chunk)
    t_data = timed(lambda: one_chunk(attend=False))          # quantise + dequantise + compress alone
    G = HQ // HKV
    tri = chunk * (chunk + 1) / 2
    flops = sum(4 * D * G * (nf * (chunk * past + tri) + (HKV - nf) * (chunk * W + tri)) for nf in counts)
    res = {"context": ctx, "chunk": chunk, "past": past, "layers": len(counts), "chunk_ms": t_all * 1e3,
           "tok_s": chunk / t_all, "quantise_dequantise_compress_ms": t_data * 1e3,
           "attention_TFLOPs": flops / max(t_all - t_data, 1e-9) / 1e12,
           "what": "put (quantise) + get (dequantise all pools to fp16) + fp16 MFMA prefill + compress, 32 layers, last chunk"}
    del cache
    torch.cuda.empty_cache()
    return res


def cpu_cfg1_end_to_end(n_layers):
    """BASELINE configs[0] end to end on the host: a random-init Llama-2-7B-32K-shape HuggingFace model (MHA 32 heads,
    linear RoPE factor 8), `enable_duo_attention_eval` (the reference's tuple-cache entry point) at 25 % retrieval heads,
    the oracle as device backend, one 4096-token prompt + 4 decode steps, all on the CPU.  `n_layers` < 32 builds a
    shallower model of the same width and scales the figure by 32 / n_layers (labelled)."""
    import numpy as np
    from transformers import LlamaConfig, LlamaForCausalLM

    from duo_attn import backend
    from duo_attn.patch import enable_duo_attention_eval
    from oracle.duo_oracle import OracleBackend

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=n_layers, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=32000, max_position_embeddings=32768, rope_theta=10000.0,
                      rope_scaling={"rope_type": "linear", "factor": 8.0}, attn_implementation="eager",
                      tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    rng = np.random.RandomState(42)
    heads = (rng.rand(n_layers, 32) < 0.25).astype(float)
    enable_duo_attention_eval(model, heads, 128, 256)
    backend._set_backend_for_testing(OracleBackend())
    try:
        ids = torch.randint(0, 32000, (1, 4096), generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            t0 = time.perf_counter()
            out = model(input_ids=ids, past_key_values=None, use_cache=True)
            t_pre = time.perf_counter() - t0
            past, tok = out.past_key_values, out.logits[:, -1:].argmax(-1)
            n_dec = 4 if n_layers == 32 else 1      # (a bf16 M = 1 linear on the host is slow: ~5 s per layer and token)
            t0 = time.perf_counter()
            for _ in range(n_dec):
                out = model(input_ids=tok, past_key_values=past, use_cache=True)
                past, tok = out.past_key_values, out.logits[:, -1:].argmax(-1)
            t_dec = (time.perf_counter() - t0) / n_dec
    finally:
        backend._set_backend_for_testing(None)
    k = 32 / n_layers
    return {"what": "Llama-2-7B-32K shape (random init, bf16) through enable_duo_attention_eval on the host, oracle backend, "
                    f"25% retrieval heads, sink 128 recent 256: 4096-token prompt + decode steps; {n_layers} of 32 layers"
                    + ("" if n_layers == 32 else f" built and timed, x{k:g}"),
            "cores": cores, "layers_timed": n_layers, "prefill_s": t_pre * k, "prefill_tok_s": 4096 / (t_pre * k),
            "decode_ms_per_token": t_dec * k * 1e3}


def model_level(args):
    """The reference's benchmark_static protocol on the whole random-init HF model of the same shape (GEMMs included):
    tools/benchmark_static.py, prefill 1 warm + 1 timed pass, decode through the captured HIP graph."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("duo_benchmark_static", os.path.join(ROOT, "tools", "benchmark_static.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = m.parse(["--max_length", str(args.ctx), "--prefilling_chunk_size", str(args.chunk), "--prefill_steps", "1",
                 "--prefill_warmup", "1", "--decode_steps", "50", "--decode_warmup", "10", "--graph", "--also_module_by_module",
                 "--all_decode_modes", "--also_tuple"])
    r = m.run(a, quiet=True)
    t = r.get("tuple_path") or {}
    return {"prefill_tok_s": r["prefill_tok_s"], "decode_ms_per_token": r["avg_generation_time_ms"],
            "decode_ms_per_token_module_by_module": r.get("avg_generation_time_module_by_module_ms"),
            # the reference's decode loop (eval/efficiency/benchmark_static.py:96-105) UNCHANGED, as this package runs it by
            # default (the step behind model(...) is captured on the way and replayed), and with every step issued from Python
            "decode_ms_per_token_reference_loop": r.get("avg_generation_time_reference_loop_ms"),
            "decode_ms_per_token_eager": r.get("avg_generation_time_eager_ms"),
            # the same model through enable_duo_attention_eval (tuple cache: README quick-start, NIAH, LongBench)
            "tuple_decode_ms_per_token": t.get("avg_generation_time_ms"), "tuple_prefill_tok_s": t.get("prefill_tok_s"),
            "decode_mode": r["decode_mode"], "kv_cache_MB": r["kv_cache_memory_MB"], "sparsity": r["sparsity"],
            "what": "whole HF Llama-3-8B-shape model, random init (tools/benchmark_static.py --graph --all_decode_modes "
                    "--also_tuple): prefill GEMMs are hipBLASLt; the decode step's token-row linears are duo_token_linear_bf16 "
                    "(module_by_module: the same step through the library GEMMs at M = 1 and separate norm / activation / "
                    "add kernels); decode_ms_per_token = explicit graph replay with evict_last inside the graph"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=131072)
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--row-block", type=int, default=-1,
                    help="N > 1: query rows per pipeline item (a chunk is handed through the stages in row blocks); "
                         "0 = whole chunks, -1 = automatic (4096 rows on up to 4 GPUs, 2048 on more)")
    ap.add_argument("--block-streams", type=int, default=1, choices=[1, 2, 3],
                    help="one GPU only (measurement): consecutive row blocks (DUO_BENCH_FORCE_BLOCKS=1) or whole chunks on alternating HIP streams")
    ap.add_argument("--virtual-stages", type=int, default=1, choices=[1, 2],
                    help="N > 1: layer blocks per rank.  2 = duo_attn.pipeline.InterleavedLayerPipeline (rank r owns blocks r and "
                         "N + r, an item goes round the ranks twice): better balance of the ragged layers, half the fill — "
                         "gloo-tested, never run on RCCL, so opt-in")
    ap.add_argument("--no-full-baseline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 FETCH_SIZE child run")
    ap.add_argument("--no-model-level", action="store_true", help="skip the whole-HF-model run (tools/benchmark_static.py)")
    ap.add_argument("--no-parity", action="store_true", help="skip the live parity check")
    ap.add_argument("--no-int4", action="store_true", help="skip the INT4-KV leg (BASELINE configs[4] kernels)")
    ap.add_argument("--no-token-linear", action="store_true",
                    help="skip the token-row-linear leg (roofline_token_linear: the decode step's q|k|v, o_proj, gate|up, down_proj)")
    ap.add_argument("--pattern", default="llama3-8b-1048k@0.5", choices=sorted(PATTERNS),
                    help="per-layer retrieval-head counts of the job (default = BASELINE configs[1])")
    ap.add_argument("--cpu-cfg1-layers", type=int, default=1,
                    help="layers of the Llama-2-7B-shape model of the host end-to-end run (cpu_baseline.cfg1_end_to_end); "
                         "32 = the whole model (minutes), 0 = skip")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    # stdout carries the ONE JSON line and nothing else: whatever a leg prints on the way (the patch API announces itself
    # like the reference's does: "Enabling DuoAttention evaluation ...") goes to stderr
    # — at the file-descriptor level too: native libraries write to fd 1 directly (gloo reports its rendezvous there)
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    from duo_attn import launch

    if args.gpus > 1 and not launch.launched_by_torchrun() and not args.traffic_probe:
        # plain `python bench.py --gpus N`: no rank environment -> start the N ranks here (one per GPU, RCCL; the
        # driver's own `python -m torch.distributed.run ... bench.py --gpus N` arrives with one and skips this)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
        os.dup2(json_out.fileno(), 1)        # the ranks inherit the real stdout; rank 0 prints the line
        raise SystemExit(launch.self_launch(__file__, sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node "
                         f"{args.gpus}, or without any launcher (bench.py then starts its own ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if world > 1:
        launch.check_visible_gpus(world)
    # DUO_BENCH_DEBUG_SHARED_GPU=1: rehearsal of the N > 1 code path on a ONE-GPU box — every rank
    # computes on cuda:0 and the hand-off goes through gloo/host memory.  Not a measurement mode.
    shared = os.environ.get("DUO_BENCH_DEBUG_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    handoff = torch.device("cpu") if shared else device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver stack
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    if args.traffic_probe:
        return traffic_probe(args, device)

    from duo_attn.pipeline import InterleavedLayerPipeline, LayerPipeline

    counts = PATTERNS[args.pattern][0][: args.layers]
    L = len(counts)
    # stage boundaries: contiguous layers, split so that the most expensive stage is as cheap as possible
    # (per-layer cost = its algorithmic prefill FLOPs — the ragged retrieval-head counts make the
    # reference's even split lopsided: 6 vs 21 retrieval heads in the first/last 4-layer block)
    pf = prefill_flops(counts, args.ctx, args.chunk)
    layer_cost = [sum(row[l] for row in pf) for l in range(L)]
    if args.virtual_stages == 2 and world > 1:
        pipe = InterleavedLayerPipeline(L, rank=rank, world_size=world, layer_costs=layer_cost)
        lr_blocks = list(pipe.blocks)
    else:
        pipe = LayerPipeline(L, rank=rank, world_size=world, layer_costs=layer_cost if world > 1 else None)
        lr_blocks = [(pipe.first_layer, pipe.last_layer)]
    lr = (lr_blocks[0][0], lr_blocks[-1][1])            # (reporting only when there are two blocks)
    local_counts = [c for lo, hi in lr_blocks for c in counts[lo:hi]]
    n_tok = args.ctx + args.decode_tokens

    def timed(hp, steps, warmup):
        for _ in range(warmup):
            run_job(hp, pipe, args.decode_tokens, world, device, handoff)
        sync_all(world)
        t0 = time.perf_counter()
        parts, locs = [], []
        for _ in range(steps):
            parts.append(run_job(hp, pipe, args.decode_tokens, world, device, handoff))
            locs.append(run_job.local)
        sync_all(world)
        t = torch.tensor([time.perf_counter() - t0, sum(p[1] for p in parts), sum(p[2] for p in parts)],
                         device=handoff, dtype=torch.float64)
        each = torch.tensor([[p[1], p[2]] for p in parts], device=handoff, dtype=torch.float64)   # per job: prefill, decode
        timed.local_jobs = [list(x) for x in locs]      # this rank's own clock: its last item done, not the barrier
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(each, op=dist.ReduceOp.MAX)
        timed.last_jobs = each.tolist()
        return (t / steps).tolist()

    hp = HotPath(counts, lr_blocks if len(lr_blocks) > 1 else lr, args.ctx, args.chunk, device)
    hp_counts_cost = sum(sum(layer_cost[lo:hi]) for lo, hi in lr_blocks)
    # (DUO_BENCH_FORCE_BLOCKS=1: row blocks on one GPU too — measures what the finer launches cost)
    if args.row_block < 0:
        # finer blocks fill the pipeline sooner but run the kernels at smaller launches (one GPU, whole job:
        # 4096-row blocks 94 %, 2048 89 %, 1024 78 % of the whole-chunk prefill rate): with P stages the
        # fill costs ~(P-1) block slots, so more stages want smaller blocks
        # (tools/pipeline_model.py on this build's one-GPU block efficiencies — 4096 rows 0.935, 2048 rows 0.872: 4096-row
        #  blocks win on 2 and, by ~2 %, on 4 stages; 2048 on 8, where the fill of the longer pipeline weighs more)
        args.row_block = 4096 if world <= 4 else 2048
    use_blocks = args.row_block > 0 and (world > 1 or os.environ.get("DUO_BENCH_FORCE_BLOCKS") == "1")
    if use_blocks:
        hp.set_row_blocks(args.row_block)
    if args.block_streams > 1:
        if world > 1:
            raise SystemExit("--block-streams is a one-GPU measurement")
        hp.set_block_streams(args.block_streams)
    t_job, t_pre, t_dec = timed(hp, args.steps, args.warmup)
    duo_jobs, duo_local = timed.last_jobs, timed.local_jobs

    # HBM-side traffic per launch is a PMC measurement (rocprofv3 --pmc FETCH_SIZE, its own pass, x2 gfx950
    # correction, MI355X_MICROARCH.md): taken NOW by a child run of this file under rocprofv3 (one prefill pass +
    # a few decode steps), never pasted from an older build's profile; null with the reason when that fails
    traffic, traffic_source = {}, "not measured (--no-traffic)"
    if world == 1 and not args.no_traffic and not args.no_kernel_roofline:
        traffic, traffic_source = measure_traffic(args)
    roof = roof_dec = None
    if not args.no_kernel_roofline:
        pre, dec = kernel_rooflines(hp, local_counts)
        tp, td = pre["flops"] / pre["seconds"], dec["bytes"] / dec["seconds"]
        roof = {"kernel": "duo_prefill_w64_kernel (4 waves x 64 rows: every prefill launch, whole chunks, key-range-split launches and row blocks alike)",
                "bound": "mfma", "achieved": tp / 1e12, "peak": MFMA_BF16_PEAK / 1e12,
                "unit": "TFLOP/s", "frac": tp / MFMA_BF16_PEAK,
                "traffic": traffic.get("duo_prefill"), "traffic_source": traffic_source,
                "avg_launch_ms": pre["seconds"] / pre["launches"] * 1e3, "launches": pre["launches"],
                "algorithmic_flops_per_launch": pre["flops"] / pre["launches"]}
        step_bytes = float(sum(decode_bytes(local_counts, args.ctx)))
        t_tok = t_dec / args.decode_tokens
        try:
            graph_dec = decode_graph_leg(hp, local_counts) if world == 1 else None
        except Exception as e:      # extra information: never at the price of the bench line
            graph_dec = {"error": f"{type(e).__name__}: {e}"}
        roof_dec = {"kernel": "duo_decode_scan_kernel", "bound": "hbm", "achieved": td / 1e9, "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": td / HBM_PEAK,
                    "traffic": traffic.get("duo_decode_scan_kernel"), "traffic_source": traffic_source,
                    "avg_launch_ms": dec["seconds"] / dec["launches"] * 1e3, "launches": dec["launches"],
                    "algorithmic_bytes_per_launch": dec["bytes"] / dec["launches"],
                    # the whole decode step of the timed job (per layer: scan with RoPE + append folded in, then the merge launch)
                    "whole_step": {"algorithmic_bytes_per_token": step_bytes, "ms_per_token": t_tok * 1e3,
                                   "achieved": step_bytes / t_tok / 1e9, "frac": step_bytes / t_tok / HBM_PEAK,
                                   "launches_per_token": 2 * len(local_counts)},
                    "captured_step": graph_dec}
    def all_ranks_sum(x):
        t = torch.tensor([float(x)], device=handoff, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    kv_bytes = all_ranks_sum(hp.cache.memory_usage)   # every rank owns the pools of its layers
    hp.free()

    full = None
    if not args.no_full_baseline:
        hpf = HotPath([HKV] * L, lr_blocks if len(lr_blocks) > 1 else lr, args.ctx, args.chunk, device)
        if use_blocks:
            hpf.set_row_blocks(args.row_block)
        # same protocol as the duo job (warm-up, then timed jobs): the denominator of both speed-ups
        fsteps, fwarm = max(1, min(args.steps, 3)), min(args.warmup, 1)
        f_job, f_pre, f_dec = timed(hpf, fsteps, fwarm)
        fj = timed.last_jobs
        full = {"job_tok_s": n_tok / f_job, "prefill_tok_s": args.ctx / f_pre,
                "decode_tok_s": args.decode_tokens / f_dec, "kv_cache_bytes": all_ranks_sum(hpf.cache.memory_usage),
                "steps": fsteps, "warmup": fwarm,
                "prefill_tok_s_min_max": [args.ctx / max(j[0] for j in fj), args.ctx / min(j[0] for j in fj)],
                "decode_tok_s_min_max": [args.decode_tokens / max(j[1] for j in fj), args.decode_tokens / min(j[1] for j in fj)]}
        hpf.free()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(counts, args.ctx, args.chunk, args.decode_tokens)
    parity = live_parity(device) if (world == 1 and not args.no_parity) else None
    int4 = None
    if world == 1 and not args.no_int4:
        try:
            int4 = int4_leg(device)
            d = int4["decode"]
            if (args.ctx, L) == (131072, 32):
                int4["whole_step_3p3M"] = int4_whole_step(device, counts)
                int4["prefill_chunk_pipeline"] = int4_prefill_chunk(device, counts, args.ctx, args.chunk)
            int4["roofline"] = {"kernel": d["kernel"], "bound": "hbm", "achieved": d["kernel_GBps"],
                                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": d["kernel_GBps"] * 1e9 / HBM_PEAK,
                                "traffic": traffic.get("duo_int4_decode"), "traffic_source": traffic_source,
                                "avg_launch_ms": d["kernel_ms"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"]}
        except Exception as e:      # extra information: never at the price of the bench line
            int4 = {"error": f"{type(e).__name__}: {e}"}
    tlin = None
    if world == 1 and not args.no_token_linear:
        try:
            tlin = token_linear_leg(device)
            tlin["roofline"]["traffic"] = traffic.get("duo_token_linear_kernel")     # mean over the four launch shapes, like algorithmic_bytes_per_launch
            tlin["roofline"]["traffic_source"] = traffic_source
            torch.cuda.empty_cache()
        except Exception as e:      # extra information: never at the price of the bench line
            tlin = {"error": f"{type(e).__name__}: {e}"}
    if cpu is not None and args.cpu_cfg1_layers > 0:
        try:
            cpu["cfg1_end_to_end"] = cpu_cfg1_end_to_end(args.cpu_cfg1_layers)
        except Exception as e:
            cpu["cfg1_end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
    mlevel = None
    if world == 1 and not args.no_model_level and (args.ctx, L) == (131072, 32):
        try:
            mlevel = model_level(args)
        except Exception as e:      # extra information: never at the price of the bench line
            mlevel = {"error": f"{type(e).__name__}: {e}"}
    # per-rank view of the pipeline (N > 1): stage boundaries, this rank's own prefill / decode seconds per job,
    # bytes handed to the next stage per job
    b2 = lr_blocks[1] if len(lr_blocks) > 1 else (-1, -1)
    mine = torch.tensor([lr_blocks[0][0], lr_blocks[0][1], sum(j[0] for j in duo_local) / len(duo_local), sum(j[1] for j in duo_local) / len(duo_local),
                         float(hp_counts_cost), b2[0], b2[1]], device=handoff, dtype=torch.float64)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)

    if rank == 0:
        line = {
            "metric": "prefill tok/s + decode tok/s @128K ctx, Llama-3-8B 50% streaming, 1 MI355X",
            "value": n_tok / t_job,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_job * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": (f"{PATTERNS[args.pattern][1]}, attention hot path only (op level): "
                             f"{args.ctx}-token chunked prefill (chunk {args.chunk}) + {args.decode_tokens} decode "
                             f"steps at {args.ctx} ctx, {L} layers ({sum(counts)} of {HKV * L} kv heads retrieval heads), "
                             f"sink {SINK} recent {RECENT}, B=1"),
                "pattern": args.pattern,
                "global_batch": 1,
                "seq_len": args.ctx,
                "prefill_chunk": args.chunk,
                "decode_tokens": args.decode_tokens,
                "parallelism": ((f"layer-pipeline pp{world}, {args.row_block}-row wavefront" if use_blocks
                                 else f"layer-pipeline pp{world}") + (", two layer blocks per rank" if args.virtual_stages == 2 else ""))
                               if world > 1 else ("single GPU" + (f", {args.row_block}-row blocks" if use_blocks else "") + (f", consecutive items on {args.block_streams} streams" if args.block_streams > 1 else "")),
            },
            "prefill_tok_s": args.ctx / t_pre,
            "decode_tok_s": args.decode_tokens / t_dec,
            "decode_ms_per_token": t_dec / args.decode_tokens * 1e3,
            "kv_cache_bytes": kv_bytes,
            "full_attention": full,
            "speedup_vs_full_attention": None if full is None else {
                "prefill": (args.ctx / t_pre) / full["prefill_tok_s"],
                "decode": (args.decode_tokens / t_dec) / full["decode_tok_s"],
                "kv_memory": full["kv_cache_bytes"] / kv_bytes,
            },
            "roofline": roof,
            "roofline_decode": roof_dec,
            "cpu_baseline": cpu,
            "duo_job_spread": {"prefill_tok_s_min_max": [args.ctx / max(j[0] for j in duo_jobs), args.ctx / min(j[0] for j in duo_jobs)],
                               "decode_tok_s_min_max": [args.decode_tokens / max(j[1] for j in duo_jobs),
                                                        args.decode_tokens / min(j[1] for j in duo_jobs)]},
            "decode_note": ("decode speed-up ceiling = K/V byte ratio (1.994x at exactly 50 % streaming heads: 8.615 vs "
                            "17.18 GB/token at 131072 ctx); whole step = scan + merge launch per layer, its HBM fraction is "
                            "roofline_decode.whole_step"),
            "parity_live": parity,
            "roofline_int4": int4,
            "roofline_token_linear": tlin,
            "model_level": mlevel,
            "pipeline": None if world == 1 else {
                "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "handoff_bytes_per_job": (args.ctx + args.decode_tokens) * HIDDEN * 2 if world > 1 else 0,
                "virtual_stages": args.virtual_stages if world > 1 else 1,
                "stages": [{"rank": r, "layers": [int(t[0]), int(t[1])], **({"second_block": [int(t[5]), int(t[6])]} if t[5] >= 0 else {}),
                            "prefill_s": float(t[2]), "decode_s": float(t[3]),
                            "share_of_prefill_flops": float(t[4]) / sum(layer_cost)} for r, t in enumerate(per_rank)],
            },
        }
        print(json.dumps(line), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
