"""Randomised runs of head-parallel tensor parallelism (duo_attn/tp.py) on the CPU (gloo, oracle backend, fp32 models): random
depth, kv-head count and GQA group, random retrieval patterns (incl. layers with none / all), world size 2 or 4, batch rows,
random chunking — through the four ways in:
    explicit        shard_model_for_tp(model, pattern), enabler and cache on the local pattern
    balanced        to_device(enable_tp=True, full_attention_heads=pattern) then the whole-model pattern to enabler + cache
    contiguous      to_device(enable_tp=True) without the pattern (the reference's contiguous split, utils.py:206-227)
    patched_first   enable_duo_attention_eval FIRST, to_device(enable_tp=True) second (the reference's NIAH / LongBench order),
                    tuple caches
against the single-process model: logits allclose at 2e-4 (fp32; the all-reduces change the summation order only).

    python tests/fuzz_tp_gloo.py --cases 12 [--seed 1]"""
import argparse
import multiprocessing as mp
import os
import random
import socket
import sys
import time
import traceback

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOCAB = 97


def _paths():
    for p in (ROOT, os.path.join(ROOT, "duo-attention_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def draw_case(rng):
    world = rng.choice([2, 2, 4])
    Hkv = rng.choice([2, 4] if world == 2 else [4])
    group = rng.choice([1, 2])
    L = rng.randint(1, 4)
    row = lambda: [float(rng.random() < 0.5) for _ in range(Hkv)]
    heads = [rng.choice([row(), row(), [0.0] * Hkv, [1.0] * Hkv]) for _ in range(L)]
    chunks = [rng.randint(1, 30) for _ in range(rng.randint(1, 3))] + [1] * rng.randint(0, 3)
    return dict(world=world, Hkv=Hkv, group=group, heads=heads, chunks=chunks, B=rng.choice([1, 1, 2]), sink=rng.choice([2, 4]),
                recent=rng.choice([4, 8, 30]), mode=rng.choice(["explicit", "balanced", "contiguous", "patched_first"]),
                seed=rng.randint(0, 2 ** 31 - 1))


def _tiny(c):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(c["seed"])
    Hq = c["Hkv"] * c["group"]
    cfg = LlamaConfig(hidden_size=Hq * 128, intermediate_size=256, num_hidden_layers=len(c["heads"]), num_attention_heads=Hq,
                      num_key_value_heads=c["Hkv"], head_dim=128, vocab_size=VOCAB, max_position_embeddings=2048,
                      rope_theta=10000.0, attn_implementation="eager", tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).float().eval()


def _ids(c):
    return torch.randint(0, VOCAB, (c["B"], sum(c["chunks"])), generator=torch.Generator().manual_seed(c["seed"] ^ 3))


def _run_static(c, model, heads):
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval

    enable_llama_duo_attention_static_kv_cache_eval(model, np.array(heads, dtype=np.float64).copy())
    cache = DuoAttentionStaticKVCache(model, heads, c["B"], sum(c["chunks"]) + 2, c["sink"], c["recent"])
    ids, outs, pos = _ids(c), [], 0
    with torch.no_grad():
        for n in c["chunks"]:
            outs.append(model(input_ids=ids[:, pos:pos + n], past_key_values=cache, use_cache=True).logits)
            pos += n
    return torch.cat(outs, 1)


def _run_tuple(c, model):
    ids, outs, pos, past = _ids(c), [], 0, None
    with torch.no_grad():
        for n in c["chunks"]:
            o = model(input_ids=ids[:, pos:pos + n], past_key_values=past, use_cache=True)
            past = o.past_key_values
            outs.append(o.logits)
            pos += n
    return torch.cat(outs, 1)


def reference_run(c):
    _paths()
    from duo_attn import backend
    from duo_attn.patch import enable_duo_attention_eval
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend(round_p=False))
    try:
        m = _tiny(c)
        if c["mode"] == "patched_first":
            enable_duo_attention_eval(m, np.array(c["heads"]), c["sink"], c["recent"])
            return _run_tuple(c, m).numpy()
        return _run_static(c, m, np.array(c["heads"])).numpy()
    finally:
        backend._set_backend_for_testing(None)


def _worker(rank, c, port, q):
    _paths()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    world = c["world"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from duo_attn import backend
        from duo_attn.patch import enable_duo_attention_eval, get_full_attention_heads
        from duo_attn.tp import shard_model_for_tp
        from duo_attn.utils import to_device
        from oracle.duo_oracle import OracleBackend

        backend._set_backend_for_testing(OracleBackend(round_p=False))
        model, H = _tiny(c), np.array(c["heads"])
        if c["mode"] == "explicit":
            local = shard_model_for_tp(model, H)
            assert local.shape == (len(c["heads"]), c["Hkv"] // world) and (np.diff(local, axis=1) <= 0).all()
            out = _run_static(c, model, local)
        elif c["mode"] == "patched_first":
            enable_duo_attention_eval(model, H.copy(), c["sink"], c["recent"])
            to_device(model, ["cpu"] * world, enable_tp=True)
            got = torch.stack(get_full_attention_heads(model)).float().numpy()
            assert np.array_equal(got, -np.sort(-H, axis=1)), got       # the gathered pattern is the reordered whole-model one
            out = _run_tuple(c, model)
        else:
            to_device(model, ["cpu"] * world, enable_tp=True, full_attention_heads=H if c["mode"] == "balanced" else None)
            out = _run_static(c, model, H)
        q.put((rank, out.numpy(), None))
        dist.barrier()
    except Exception:      # noqa: BLE001
        q.put((rank, "error", traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def run_case(c):
    want = reference_run(c)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, c, port, q)) for r in range(c["world"])]
    for p in procs:
        p.start()
    try:
        msgs = [q.get(timeout=400) for _ in range(c["world"])]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank, got, tb in msgs:
        assert not isinstance(got, str), f"rank {rank} raised:\n{tb}"
        assert got.shape == want.shape, (got.shape, want.shape)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-4), f"rank {rank}: max |diff| {np.abs(got - want).max():.3e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad, t0 = 0, time.time()
    for _ in range(a.cases):
        c = draw_case(rng)
        try:
            run_case(c)
            print("ok  ", c, flush=True)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print("FAIL", c, "\n    ", f"{type(e).__name__}: {str(e)[:1500]}", flush=True)
    print(f"{a.cases} cases in {time.time() - t0:.0f} s, {bad} failed (seed {a.seed})")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
