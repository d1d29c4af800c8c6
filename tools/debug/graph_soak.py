#!/usr/bin/env python
"""Soak of the (opt-in) automatic decode-step graph — VERDICT r5 item 1.

One abort (core dump) happened inside tests/test_auto_graph_gpu.py in one of nine round-5 runs and was never reproduced.
This harness hammers the path the abort must have been on, with ``faulthandler`` armed so that a fatal signal leaves a
Python backtrace of every thread on stderr (kept by the caller: tools/debug/graph_soak.sh writes it under gpurun_out/).

Per round, on a three-layer model of the test suite's geometry (reference loop: eval/efficiency/benchmark_static.py:96-105):

  a. the reference's decode loop (``model(...)`` + ``evict_last(1)``) replayed ``--replays`` times;
  b. a growing generation across length-bucket boundaries (re-capture at 64 / 128 / 256 rows);
  c. ``clear()`` + a new prompt of another length through the same cache (counters re-uploaded, maybe re-captured);
  d. a retired signature: the ``.data`` of one weight swapped to new storage -> the old graph is dropped mid-stream;
  e. the cache (and its graph) dropped WITHOUT a synchronisation while the last replay is still in flight, followed at once
     by a new cache + capture, with a ``gc.collect()`` thrown in;
  f. two caches captured, replayed alternately on two streams;
  g. a direct ``DecodeStepGraph`` whose re-capture is made to fail, then retried.

Every replayed logit is compared bit for bit with an eager twin every ``--check-every`` rounds (the eager twin costs as
much as the soak itself).  Exit code 0 = no mismatch and no fault.  ``--seconds`` bounds the wall time.
"""
import argparse
import faulthandler
import gc
import os
import sys
import time

faulthandler.enable(all_threads=True)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "duo-attention_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DEV = "cuda:0"


def tiny(seed):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128, vocab_size=211, max_position_embeddings=8192,
                      rope_theta=500000.0, attn_implementation="eager", tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).to(torch.bfloat16).eval().to(DEV)


HEADS = np.array([[1.0, 0.0], [0.0, 0.0], [1.0, 1.0]])


def setup(seed, max_size=1100, sink=16, recent=48):
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    model = tiny(seed)
    enable_llama_duo_attention_static_kv_cache_eval(model, HEADS.copy())
    return model, DuoAttentionStaticKVCache(model, HEADS, 1, max_size, sink, recent)


def new_cache(model, max_size=1100, sink=16, recent=48):
    from duo_attn.patch.llama import DuoAttentionStaticKVCache

    return DuoAttentionStaticKVCache(model, HEADS, 1, max_size, sink, recent)


class Soak:
    def __init__(self, args):
        from duo_attn import graph

        self.graph, self.args = graph, args
        self.ids = torch.randint(0, 211, (1, 4096), generator=torch.Generator().manual_seed(1)).to(DEV)
        self.model, self.kv = setup(21)
        self.twin, self.twin_kv = setup(21)         # same weights: the eager reference
        self.replays = self.captures = self.checked = self.mismatches = 0
        self.counts = {k: 0 for k in "abcdefg"}

    # -- helpers -------------------------------------------------------------------------------------------------
    def prefill(self, m, c, lo, hi):
        with torch.no_grad():
            return m(input_ids=self.ids[:, lo:hi], past_key_values=c, use_cache=True).logits[:, -1, :].argmax(-1).unsqueeze(1)

    def decode(self, m, c, tok, n, auto, evict, feed_lo=None, keep=True):
        self.graph.AUTO_DECODE_GRAPH = auto
        outs = []
        with torch.no_grad():
            for i in range(n):
                o = m(input_ids=tok, past_key_values=c, use_cache=True)
                if keep:
                    outs.append(o.logits)
                if evict:
                    c.evict_last(1)
                elif feed_lo is not None:
                    tok = self.ids[:, feed_lo + i:feed_lo + i + 1]
        if auto:
            self.replays += n
        return outs

    def same(self, got, want, what):
        self.checked += len(got)
        for s, (a, b) in enumerate(zip(got, want)):
            if not torch.equal(a, b):
                self.mismatches += 1
                print(f"MISMATCH {what} step {s}: max |d| = {(a.float() - b.float()).abs().max().item():.3e}", flush=True)

    # -- scenarios -----------------------------------------------------------------------------------------------
    def scen_a(self, check):
        """the reference loop, many replays of one graph"""
        n = self.args.replays
        for m, c in ((self.model, self.kv), (self.twin, self.twin_kv)):
            c.clear()
        t = self.prefill(self.model, self.kv, 0, 300)
        self.prefill(self.twin, self.twin_kv, 0, 300)
        got = self.decode(self.model, self.kv, t, n, True, True, keep=check)
        if check:
            self.same(got, self.decode(self.twin, self.twin_kv, t, n, False, True), "a")

    def scen_b(self, check):
        """growing generation through the 64 / 128 / 256-row bucket boundaries"""
        for m, c in ((self.model, self.kv), (self.twin, self.twin_kv)):
            c.clear()
        t = self.prefill(self.model, self.kv, 500, 558)
        self.prefill(self.twin, self.twin_kv, 500, 558)
        got = self.decode(self.model, self.kv, t, 210, True, False, feed_lo=1000, keep=check)
        if check:
            self.same(got, self.decode(self.twin, self.twin_kv, t, 210, False, False, feed_lo=1000), "b")

    def scen_c(self, check, r):
        """clear() + prompts of other lengths through the same cache"""
        for k, n in enumerate((20 + r % 37, 700 + r % 211, 90)):
            for m, c in ((self.model, self.kv), (self.twin, self.twin_kv)):
                c.clear()
            t = self.prefill(self.model, self.kv, 100 * k, 100 * k + n)
            self.prefill(self.twin, self.twin_kv, 100 * k, 100 * k + n)
            got = self.decode(self.model, self.kv, t, 6, True, bool(k & 1), feed_lo=2000, keep=check)
            if check:
                self.same(got, self.decode(self.twin, self.twin_kv, t, 6, False, bool(k & 1), feed_lo=2000), f"c{k}")

    def scen_d(self, check, r):
        """a retired signature: one weight's storage swapped while the graph's last replay may still be running"""
        gen = torch.Generator().manual_seed(100 + r)
        pick = (lambda m: m.model.layers[r % 3].self_attn.k_proj.weight, lambda m: m.lm_head.weight,
                lambda m: m.model.layers[r % 3].post_attention_layernorm.weight)[r % 3]
        new = (torch.randn(pick(self.model).shape, generator=gen) * 0.05).to(torch.bfloat16).to(DEV)
        for m, c in ((self.model, self.kv), (self.twin, self.twin_kv)):
            c.clear()
        t = self.prefill(self.model, self.kv, 0, 130)
        self.prefill(self.twin, self.twin_kv, 0, 130)
        self.decode(self.model, self.kv, t, 5, True, True, keep=False)
        for m in (self.model, self.twin):
            pick(m).data = new.clone()              # no synchronisation: the old storage goes now, the old graph at the next call
        for m, c in ((self.model, self.kv), (self.twin, self.twin_kv)):
            c.clear()
        t = self.prefill(self.model, self.kv, 0, 130)
        self.prefill(self.twin, self.twin_kv, 0, 130)
        got = self.decode(self.model, self.kv, t, 5, True, True, keep=check)
        if check:
            self.same(got, self.decode(self.twin, self.twin_kv, t, 5, False, True), "d")

    def scen_e(self, check, r):
        """the cache and its graph dropped while a replay is in flight; a new cache and capture right behind it"""
        for _ in range(3):
            kv = new_cache(self.model)
            t = self.prefill(self.model, kv, 0, 900)
            self.decode(self.model, kv, t, 4, True, True, keep=False)
            del kv                                   # replay possibly still executing
            if r & 1:
                gc.collect()
        kv = new_cache(self.model)
        t = self.prefill(self.model, kv, 0, 200)
        got = self.decode(self.model, kv, t, 4, True, True, keep=check)
        if check:
            self.twin_kv.clear()
            self.prefill(self.twin, self.twin_kv, 0, 200)
            self.same(got, self.decode(self.twin, self.twin_kv, t, 4, False, True), "e")

    def scen_f(self, check):
        """two caches, two graphs, replayed alternately on two streams"""
        kvs = [new_cache(self.model), new_cache(self.model)]
        toks = [self.prefill(self.model, kv, 0, 650 + 10 * i) for i, kv in enumerate(kvs)]
        for kv, t in zip(kvs, toks):
            self.decode(self.model, kv, t, 3, True, True, keep=False)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        self.graph.AUTO_DECODE_GRAPH = True
        got = [[], []]
        with torch.no_grad():
            for s in range(12):
                for i, (kv, st) in enumerate(zip(kvs, streams)):
                    with torch.cuda.stream(st):
                        got[i].append(self.model(input_ids=toks[i], past_key_values=kv, use_cache=True).logits)
                        kv.evict_last(1)
                    st.synchronize()
        self.replays += 24
        torch.cuda.synchronize()
        if check:
            for i in range(2):
                self.twin_kv.clear()
                self.prefill(self.twin, self.twin_kv, 0, 650 + 10 * i)
                # (the three warm-up steps first: a step + evict_last(1) on a saturated streaming window leaves it one row
                #  short, so the first step after a prefill is not the steady state)
                self.same(got[i], self.decode(self.twin, self.twin_kv, toks[i], 15, False, True)[3:], f"f{i}")

    def scen_g(self, check):
        """a direct DecodeStepGraph: the re-capture for the next bucket fails once (simulated), is retried, succeeds"""
        from duo_attn.graph import DecodeStepGraph, RecaptureError

        kv = new_cache(self.model)
        t = self.prefill(self.model, kv, 0, 62)
        tok = t.clone()
        with torch.no_grad():
            for _ in range(2):
                self.model(input_ids=tok, past_key_values=kv, use_cache=True, _duo_no_auto_graph=True)
                kv.evict_last(1)

        def step():
            with torch.no_grad():
                return self.model(input_ids=tok, past_key_values=kv, use_cache=True, _duo_no_auto_graph=True).logits

        g = DecodeStepGraph(kv, step, evict_after=0)
        outs = [g.replay().clone() for _ in range(2)]           # 62 -> 64 rows
        orig, fail = g._body, [True]

        def body():
            if fail[0]:
                fail[0] = False
                raise RuntimeError("simulated capture failure")
            return orig()

        g._body = body
        try:
            g.replay()                                          # 65 rows: new bucket, capture fails
            raise AssertionError("the simulated failure did not surface")
        except RecaptureError:
            pass
        assert g.plan_key is None
        outs += [g.replay().clone() for _ in range(3)]          # retried, captured, replayed
        assert g.captures == 2 and g.plan_key is not None
        self.replays += 5
        if check:
            self.twin_kv.clear()
            self.prefill(self.twin, self.twin_kv, 0, 62)
            self.graph.AUTO_DECODE_GRAPH = False
            want = []
            with torch.no_grad():
                for _ in range(5):
                    want.append(self.twin(input_ids=t, past_key_values=self.twin_kv, use_cache=True).logits)
            self.same(outs, want, "g")

    def run(self):
        t0, r = time.time(), 0
        while time.time() - t0 < self.args.seconds and r < self.args.rounds:
            check = r % self.args.check_every == 0
            for name, fn in (("a", lambda: self.scen_a(check)), ("b", lambda: self.scen_b(check)), ("c", lambda: self.scen_c(check, r)),
                             ("d", lambda: self.scen_d(check, r)), ("e", lambda: self.scen_e(check, r)), ("f", lambda: self.scen_f(check)),
                             ("g", lambda: self.scen_g(check))):
                fn()
                self.counts[name] += 1
            r += 1
            if r % 10 == 0:
                torch.cuda.synchronize()
                print(f"round {r}: {self.replays} replays, {self.checked} logits compared, {self.mismatches} mismatches, "
                      f"{len(self.graph._retired)} graphs awaiting release, {time.time() - t0:.0f} s", flush=True)
        torch.cuda.synchronize()
        self.graph._drain_retired(block=True)
        print(f"SOAK DONE rounds={r} replays={self.replays} compared={self.checked} mismatches={self.mismatches} "
              f"scenarios={self.counts} wall={time.time() - t0:.0f}s", flush=True)
        return 1 if self.mismatches else 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--rounds", type=int, default=10 ** 9)
    ap.add_argument("--replays", type=int, default=400, help="replays of scenario a per round")
    ap.add_argument("--check-every", type=int, default=4)
    sys.exit(Soak(ap.parse_args()).run())
