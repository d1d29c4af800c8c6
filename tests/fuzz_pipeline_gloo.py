"""Randomised runs of the model-level layer pipeline on the CPU (gloo, oracle backend): tiny random-init Llama models of random
depth and head pattern, world size 2 or 3, batch rows, prompt / chunk / row-block sizes that do and do not divide each other,
through both entry points — the reference harness's drop-in loop (``to_device(enable_pp=True)``) and ``PipelinedCausalLM`` —
against the single-process run of the same model: greedy tokens equal, logits equal (row blocks: within 2e-2, the CPU GEMM
shapes change).  Reference: duo_attn/utils.py:228-283 (pipeline placement), eval/efficiency/benchmark_static.py:68-105.

    python tests/fuzz_pipeline_gloo.py --cases 12 [--seed 1]"""
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
VOCAB = 199


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "duo-attention_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def draw_case(rng):
    world = rng.choice([2, 2, 3])
    L = rng.randint(world, 6)
    Hkv = rng.choice([1, 2])
    group = rng.choice([1, 2])
    heads = [[float(rng.random() < 0.5) for _ in range(Hkv)] for _ in range(L)]
    chunk = rng.choice([8, 16, 31, 32, 50])
    return dict(world=world, heads=heads, Hkv=Hkv, group=group, B=rng.choice([1, 1, 2]), sink=rng.choice([2, 4]),
                recent=rng.choice([6, 12, 40]), prompt=rng.randint(5, 90), chunk=chunk, n_new=rng.randint(1, 5),
                mode=rng.choice(["drop_in", "chunks", "row_blocks"]), row_block=rng.choice([4, 8, 13]), seed=rng.randint(0, 2 ** 31 - 1))


def _tiny(c):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(c["seed"])
    Hq = c["Hkv"] * c["group"]
    cfg = LlamaConfig(hidden_size=Hq * 128, intermediate_size=256, num_hidden_layers=len(c["heads"]), num_attention_heads=Hq,
                      num_key_value_heads=c["Hkv"], head_dim=128, vocab_size=VOCAB, max_position_embeddings=4096,
                      rope_theta=10000.0, attn_implementation="eager", tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).to(torch.bfloat16).eval()


def _ids(c):
    return torch.randint(0, VOCAB, (c["B"], c["prompt"]), generator=torch.Generator().manual_seed(c["seed"] ^ 9))


def reference_run(c):
    _setup_paths()
    from duo_attn import backend
    from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend())
    try:
        model = _tiny(c)
        enable_llama_duo_attention_static_kv_cache_eval(model, np.array(c["heads"]))
        kv = DuoAttentionStaticKVCache(model, c["heads"], c["B"], c["prompt"] + c["n_new"] + 2, c["sink"], c["recent"])
        ids = _ids(c)
        with torch.no_grad():
            for i in range(0, c["prompt"], c["chunk"]):
                out = model(input_ids=ids[:, i:i + c["chunk"]], past_key_values=kv, use_cache=True)
            prefill_logits = out.logits[:, -1:].float().numpy()
            tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
            toks, steps = [], []
            for _ in range(c["n_new"]):
                out = model(input_ids=tok, past_key_values=kv, use_cache=True)
                steps.append(out.logits.float().numpy())
                tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
                toks.append(tok[:, 0].tolist())
        return prefill_logits, toks, steps
    finally:
        backend._set_backend_for_testing(None)


def _worker(rank, c, port, q):
    _setup_paths()
    from helpers import P2PAudit

    # (how every point-to-point call was issued is recorded and compared hop by hop at the end: helpers.check_p2p_logs)
    with P2PAudit() as audit:
        _worker_body(rank, c, port, q, audit)


def _worker_body(rank, c, port, q, audit):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    world = c["world"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from duo_attn import backend
        from duo_attn.patch.llama import DuoAttentionStaticKVCache, enable_llama_duo_attention_static_kv_cache_eval
        from duo_attn.pipeline import PipelinedCausalLM
        from duo_attn.utils import to_device
        from oracle.duo_oracle import OracleBackend

        backend._set_backend_for_testing(OracleBackend())
        model, ids, B = _tiny(c), _ids(c), c["B"]
        cap = c["prompt"] + c["n_new"] + 2
        if c["mode"] == "drop_in":
            model = to_device(model, ["cpu"] * world, enable_pp=True)
            enable_llama_duo_attention_static_kv_cache_eval(model, np.array(c["heads"]))
            kv = DuoAttentionStaticKVCache(model, c["heads"], B, cap, c["sink"], c["recent"])
            with torch.no_grad():
                for i in range(0, c["prompt"], c["chunk"]):
                    last = i + c["chunk"] >= c["prompt"]
                    out = model(input_ids=ids[:, i:i + c["chunk"]], past_key_values=kv, use_cache=True, sync_logits=last)
                prefill_logits = out.logits[:, -1:].float().numpy()
                tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
                toks, steps = [], []
                for _ in range(c["n_new"]):
                    out = model(input_ids=tok, past_key_values=kv, use_cache=True)
                    steps.append(out.logits.float().numpy())
                    tok = out.logits[:, -1, :].argmax(-1, keepdim=True)
                    toks.append(tok[:, 0].tolist())
        else:
            enable_llama_duo_attention_static_kv_cache_eval(model, np.array(c["heads"]))
            pl = PipelinedCausalLM(model, c["heads"], "cpu")
            kv = pl.make_kv_cache(B, cap, c["sink"], c["recent"])
            logits = pl.prefill(ids, kv, c["chunk"], row_block=c["row_block"] if c["mode"] == "row_blocks" else None)
            logits = pl.pp.broadcast_from_last(logits, (B, 1, VOCAB), torch.bfloat16)
            prefill_logits = logits.float().numpy()
            tok = logits[:, -1, :].argmax(-1, keepdim=True)
            out, lg = pl.decode(tok, kv, c["n_new"], return_logits=True)
            toks = [out[:, i].tolist() for i in range(out.shape[1])]
            steps = [x.float().numpy() for x in lg]
            assert kv.kv_seq_len == c["prompt"] + c["n_new"]
        from helpers import check_p2p_logs

        logs = [None] * world
        dist.all_gather_object(logs, list(audit.calls))
        check_p2p_logs(logs)
        q.put((rank, prefill_logits, toks, steps))
        dist.barrier()
    except Exception:      # noqa: BLE001
        q.put((rank, "error", traceback.format_exc(), None))
        raise
    finally:
        dist.destroy_process_group()


def run_case(c):
    exp_logits, exp_toks, exp_steps = reference_run(c)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
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
    for m in msgs:
        assert not (isinstance(m[1], str) and m[1] == "error"), f"rank {m[0]} raised:\n{m[2]}"
    last = [m for m in msgs if m[0] == c["world"] - 1][0]
    _, got_logits, got_toks, got_steps = last
    loose = c["mode"] == "row_blocks"
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-9))
    if loose:
        assert rel(got_logits, exp_logits) < 2e-2, f"prefill logits rel {rel(got_logits, exp_logits):.3e}"
    else:
        assert np.array_equal(got_logits, exp_logits), f"prefill logits differ (rel {rel(got_logits, exp_logits):.3e})"
        assert got_toks == exp_toks, (got_toks, exp_toks)
        for a, b in zip(got_steps, exp_steps):
            assert np.array_equal(a, b), f"decode logits differ (rel {rel(a, b):.3e})"
    if c["mode"] == "drop_in":
        for m in msgs:      # S == 1: the logits reach every rank, every rank decodes the same tokens
            assert m[2] == got_toks, f"rank {m[0]} decoded {m[2]}, the last stage {got_toks}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad, t0 = 0, time.time()
    for i in range(a.cases):
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
