"""`python bench.py --gpus N` (and tools/benchmark_static.py --pp/--tp --gpus N) without a launcher start their own ranks.

VERDICT r3 item 1: the driver's N = 1 command shape, `python bench.py --gpus 8`, used to SystemExit because no
torch.distributed.run environment was present.  CPU legs: the launcher plumbing itself — command line, rendezvous on
127.0.0.1, a real two-rank gloo job re-executed through ``duo_attn.launch.self_launch``, the GPU-count check.  GPU leg:
plain ``python bench.py --gpus 2`` on the one-GPU box in the shared-GPU rehearsal mode, JSON line parsed."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_torchrun_command_line():
    from duo_attn import launch

    cmd = launch.torchrun_command("bench.py", ["--gpus", "4", "--steps", "2"], 4, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.abspath("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "2"]


def test_rank_environment_detection_and_gpu_count_check():
    from duo_attn import launch

    assert not launch.launched_by_torchrun({})
    assert not launch.launched_by_torchrun({"WORLD_SIZE": "2"})
    assert launch.launched_by_torchrun({"WORLD_SIZE": "2", "RANK": "1"})
    launch.check_visible_gpus(2, visible=2, env={})
    launch.check_visible_gpus(8, visible=1, env={launch.SHARED_GPU_ENV: "1"})
    with pytest.raises(SystemExit) as e:
        launch.check_visible_gpus(8, visible=1, env={})
    assert "--gpus 8 but 1 GPU(s) visible" in str(e.value) and launch.SHARED_GPU_ENV in str(e.value)


def test_self_launch_runs_a_two_rank_job(tmp_path):
    """a script that finds no rank environment re-executes itself as two gloo ranks; rank 0 prints the one result line"""
    script = tmp_path / "job.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys
        sys.path.insert(0, {os.path.join(ROOT, "duo-attention_amd")!r})
        from duo_attn import launch
        if not launch.launched_by_torchrun():
            raise SystemExit(launch.self_launch(__file__, sys.argv[1:], 2, visible=2))
        import torch, torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({{"world": dist.get_world_size(), "sum": t.item(), "argv": sys.argv[1:],
                              "master": os.environ["MASTER_ADDR"]}}), flush=True)
        dist.destroy_process_group()
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script), "--flag", "7"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res == {"world": 2, "sum": 3.0, "argv": ["--flag", "7"], "master": "127.0.0.1"}
    assert "starting 2 ranks" in r.stderr


def test_bench_and_benchmark_static_take_the_launcher_path():
    """both entry points consult duo_attn.launch before they look at WORLD_SIZE"""
    b = open(os.path.join(ROOT, "bench.py")).read()
    main = b[b.index("def main():"):]
    assert main.index("launch.self_launch(__file__, sys.argv[1:], args.gpus)") < main.index('int(os.environ.get("WORLD_SIZE", "1"))')
    s = open(os.path.join(ROOT, "tools", "benchmark_static.py")).read()
    assert "launch.self_launch(__file__, sys.argv[1:], n)" in s and "launch.check_visible_gpus(world)" in s


def _run_bench(extra_env, *args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


SMALL = ["--steps", "1", "--warmup", "0", "--ctx", "8192", "--chunk", "4096", "--layers", "4", "--decode-tokens", "4",
         "--row-block", "2048", "--no-cpu-baseline", "--no-traffic", "--no-model-level", "--no-int4", "--no-token-linear",
         "--no-parity"]


@pytest.mark.gpu
def test_plain_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher, on the one-GPU box in the shared-GPU rehearsal mode: the two ranks are
    started by bench.py itself, stdout is the ONE JSON line, and it describes a two-stage pipeline"""
    r = _run_bench({"DUO_BENCH_DEBUG_SHARED_GPU": "1"}, "--gpus", "2", *SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    pipe = line["pipeline"]
    assert pipe["world_size"] == 2 and pipe["backend"] == "gloo"          # (RCCL: "nccl" when every rank has its own GPU)
    assert [s["rank"] for s in pipe["stages"]] == [0, 1]
    assert pipe["stages"][0]["layers"][1] == pipe["stages"][1]["layers"][0] and pipe["stages"][1]["layers"][1] == 4
    assert line["config"]["parallelism"].startswith("layer-pipeline pp2")
    assert line["roofline"]["frac"] > 0 and line["full_attention"]["job_tok_s"] > 0


@pytest.mark.gpu
def test_bench_gpus_2_with_two_layer_blocks_per_rank():
    """`python bench.py --gpus 2 --virtual-stages 2` (shared-GPU rehearsal): rank r owns layer blocks r and 2 + r, every item goes
    round the two ranks twice (duo_attn.pipeline.InterleavedLayerPipeline) — the N > 1 code path of the opt-in form end to
    end on the HIP kernels, one JSON line, both blocks of every rank reported, every layer owned exactly once"""
    r = _run_bench({"DUO_BENCH_DEBUG_SHARED_GPU": "1"}, "--gpus", "2", "--virtual-stages", "2", *SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout
    line = json.loads(out[0])
    pipe = line["pipeline"]
    assert line["n_gpus"] == 2 and line["value"] > 0 and pipe["virtual_stages"] == 2
    blocks = sorted([tuple(s["layers"]) for s in pipe["stages"]] + [tuple(s["second_block"]) for s in pipe["stages"]])
    assert len(blocks) == 4 and blocks[0][0] == 0 and blocks[-1][1] == 4 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    # rank r owns the r-th and the (2 + r)-th block in layer order
    for s in pipe["stages"]:
        assert blocks.index(tuple(s["layers"])) == s["rank"] and blocks.index(tuple(s["second_block"])) == 2 + s["rank"]
    assert "two layer blocks per rank" in line["config"]["parallelism"]
    assert abs(sum(s["share_of_prefill_flops"] for s in pipe["stages"]) - 1.0) < 1e-6


@pytest.mark.gpu
def test_plain_bench_gpus_2_on_one_gpu_is_a_clear_error():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _run_bench({"DUO_BENCH_DEBUG_SHARED_GPU": "0"}, "--gpus", "2", *SMALL, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but 1 GPU(s) visible" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_single_gpu_path_is_unchanged():
    """N = 1: no launcher involved, one JSON line, no pipeline section"""
    r = _run_bench({}, "--gpus", "1", *SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert line["n_gpus"] == 1 and line["pipeline"] is None and "starting" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["--pp", "--tp"])
def test_benchmark_static_starts_its_own_ranks(mode):
    """`python tools/benchmark_static.py --pp|--tp --gpus 2` with no launcher (shared-GPU rehearsal: both ranks on cuda:0, gloo):
    the reference's benchmark protocol on the sharded random-init model, one JSON result from rank 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DUO_BENCH_DEBUG_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "benchmark_static.py"), mode, "--gpus", "2",
                        "--shape", "mistral-7b-v0.2", "--max_length", "3001", "--prefilling_chunk_size", "1024",
                        "--prefill_steps", "1", "--prefill_warmup", "0", "--decode_steps", "3", "--decode_warmup", "1"]
                       + (["--row_block", "512"] if mode == "--pp" else []),
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "2 ranks (gloo" in res["mode"] and res["prefill_tok_s"] > 0 and res["avg_generation_time_ms"] > 0
    assert len(res["stages" if mode == "--pp" else "ranks"]) == 2
    assert "starting 2 ranks" in r.stderr


# ----------------------------------------------------------------------------------------------------------------------
# round 6: to_device(model, [gpu ids], enable_tp=True) inside ONE python process — how the reference's
# scripts/niah.sh and scripts/longbench.sh start their harnesses — starts its own ranks
# ----------------------------------------------------------------------------------------------------------------------
def test_original_command_line_is_recovered_for_the_relaunch():
    from duo_attn import launch

    py = sys.executable
    assert launch.original_command([py, "eval/needle/needle_in_haystack.py", "-s", "300"]) == \
        [os.path.abspath("eval/needle/needle_in_haystack.py"), "-s", "300"]
    assert launch.original_command([py, "-u", "-X", "faulthandler", "-W", "ignore", "x.py", "-m", "keep"]) == \
        [os.path.abspath("x.py"), "-m", "keep"]
    assert launch.original_command([py, "-m", "pkg.tool", "--flag"]) == ["-m", "pkg.tool", "--flag"]
    for bad in ([py, "-c", "print(1)"], [py], [py, "-"]):
        with pytest.raises(ValueError, match="torch.distributed.run"):
            launch.original_command(bad)


def test_to_device_with_a_device_list_starts_its_own_ranks(tmp_path):
    """a plain ``python script.py`` that calls ``to_device(model, [0, 1], enable_tp=True)`` with no process group and no rank
    environment (reference eval/needle/needle_in_haystack.py:213-214): the command line is started again as two ranks, the
    ranks initialise their group inside to_device (gloo: no GPU here), rank 0 alone prints and writes, and its result file
    is never visible half-written; the first process exits with the ranks' exit code"""
    script = tmp_path / "harness.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys
        for p in ({ROOT!r}, {os.path.join(ROOT, "duo-attention_amd")!r}, {os.path.join(ROOT, "tests")!r}):
            sys.path.insert(0, p)
        import numpy as np, torch
        import torch.distributed as dist
        from transformers import LlamaConfig, LlamaForCausalLM
        from duo_attn import backend
        from duo_attn.patch import enable_duo_attention_eval
        from duo_attn.utils import to_device
        from oracle.duo_oracle import OracleBackend
        backend._set_backend_for_testing(OracleBackend())
        orig_to = torch.nn.Module.to
        torch.nn.Module.to = lambda self, *a, **k: orig_to(self, *[("cpu" if isinstance(x, str) and x.startswith("cuda") else x) for x in a], **k)
        torch.cuda.set_device = lambda *a, **k: None
        torch.manual_seed(3)
        cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          head_dim=128, vocab_size=64, max_position_embeddings=512, rope_theta=10000.0, tie_word_embeddings=False)
        model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
        print("loading done, world:", os.environ.get("WORLD_SIZE", "none"), flush=True)
        enable_duo_attention_eval(model, np.array([[1.0, 0.0], [0.0, 1.0]]), 4, 12)       # the harness's order: patch, then shard
        model = to_device(model, [0, 1], enable_tp=True)
        assert dist.is_initialized() and dist.get_world_size() == 2
        ids = torch.randint(0, 64, (1, 40), generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            out = model(input_ids=ids[:, :30], past_key_values=None, use_cache=True)
            toks = []
            for t in range(30, 36):
                out = model(input_ids=ids[:, t:t + 1], past_key_values=out.past_key_values, use_cache=True)
                toks.append(int(out.logits[0, -1].argmax()))
        os.makedirs(sys.argv[1], exist_ok=True)
        with open(os.path.join(sys.argv[1], "result.json"), "w") as f:
            json.dump({{"rank": dist.get_rank(), "tokens": toks}}, f)
        print("RESULT", json.dumps(toks), flush=True)
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PYTHONPATH")}
    env["DUO_BENCH_DEBUG_SHARED_GPU"] = "1"       # (no second GPU to count: the rehearsal mode skips the visible-GPU check)
    out_dir = tmp_path / "out"
    r = subprocess.run([sys.executable, str(script), str(out_dir)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "starting this command again as 2 ranks" in r.stderr
    # the first process and both ranks load the model; only rank 0 of the ranks speaks after to_device
    assert r.stdout.count("loading done, world: none") == 1 and r.stdout.count("loading done, world: 2") == 2
    assert r.stdout.count("RESULT") == 1
    res = json.load(open(out_dir / "result.json"))
    assert res["rank"] == 0 and len(res["tokens"]) == 6
    assert not [f for f in os.listdir(out_dir) if "tmp-rank0" in f]
    # ... and the tokens are the single-process model's
    single = subprocess.run([sys.executable, "-c", textwrap.dedent(f"""
        import sys
        for p in ({ROOT!r}, {os.path.join(ROOT, "duo-attention_amd")!r}):
            sys.path.insert(0, p)
        import numpy as np, torch, json
        from transformers import LlamaConfig, LlamaForCausalLM
        from duo_attn import backend
        from duo_attn.patch import enable_duo_attention_eval
        from oracle.duo_oracle import OracleBackend
        backend._set_backend_for_testing(OracleBackend())
        torch.manual_seed(3)
        cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          head_dim=128, vocab_size=64, max_position_embeddings=512, rope_theta=10000.0, tie_word_embeddings=False)
        model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
        enable_duo_attention_eval(model, np.array([[1.0, 0.0], [0.0, 1.0]]), 4, 12)
        ids = torch.randint(0, 64, (1, 40), generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            out = model(input_ids=ids[:, :30], past_key_values=None, use_cache=True)
            toks = []
            for t in range(30, 36):
                out = model(input_ids=ids[:, t:t + 1], past_key_values=out.past_key_values, use_cache=True)
                toks.append(int(out.logits[0, -1].argmax()))
        print("RESULT", json.dumps(toks))
    """)], capture_output=True, text=True, timeout=600, env=env)
    assert single.returncode == 0, single.stderr[-2000:]
    want = json.loads(single.stdout.split("RESULT", 1)[1])
    assert res["tokens"] == want
