"""Build container only: the reference's OWN efficiency harnesses run VERBATIM against this package.

``eval/efficiency/benchmark_static.py`` (the loop BASELINE.json's metric is defined on) and ``benchmark_dynamic.py`` (the
tuple-cache / ``enable_duo_attention_eval`` loop) are executed from /root/reference with ``runpy`` — not a line of them is
copied or edited — while ``import duo_attn`` resolves to THIS package: every name they import, every signature they call and
every attribute they read has to exist for the script to reach its last line and write ``benchmark_result.txt``
(tests/golden/run_reference_harness.py: tiny random Llama + tokenizer + head pattern in a temp directory; the CPU oracle as the
device backend and a "cuda" -> "cpu" shim, because this container has no GPU and the product has no CPU path).
Skipped where /root/reference does not exist (the GPU box); nothing under -m gpu, smoke() or bench.py reads the reference."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/eval/efficiency")


def _run(script, tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "run_reference_harness.py"), script, str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    return dict(x.split(": ", 1) for x in json.loads(line[-1][len("RESULT "):])), r.stdout


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_reference_benchmark_static_runs_unchanged(tmp_path):
    """benchmark_static.py:20-119 — get_tokenizer / get_model / to_device / load_attn_pattern / sparsify_attention_heads /
    enable_llama_duo_attention_static_kv_cache_eval / DuoAttentionStaticKVCache(model, heads, 1, max_size, sink, recent) /
    13 chunked-prefill passes with kv_cache.clear() / 150 decode calls with kv_cache.evict_last(1) / kv_cache.memory_usage"""
    res, out = _run("benchmark_static.py", tmp_path)
    assert res["Context length"] == "50" and res["Prefilling chunk size"] == "20" and res["Sparsity"] == "0.5"
    assert "True Sparsity: 0.5" in out and "torch.Size([1, 49])" in out and "Max size: 54" in out
    # three layers x (one retrieval head x 54 rows + one streaming head x 16 rows) x 128 dims x bf16 x (K, V) = 107 520 B:
    # the number the reference's own formula gives on its token-major pools (static_kv_cache.py:299-315)
    assert res["KV cache memory usage"] == "0.1025 MB"
    assert float(res["Average generation time"].split()[0]) > 0 and float(res["Average context time"].split()[0]) > 0


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_reference_benchmark_dynamic_runs_unchanged(tmp_path):
    """benchmark_dynamic.py:17-103 — enable_duo_attention_eval(model, heads, 16, 64), one single-shot prefill with
    past_key_values=None, outputs.past_key_values handed back for 110 decode calls"""
    res, out = _run("benchmark_dynamic.py", tmp_path)
    assert res["Context length"] == "50" and res["Sparsity"] == "0.5"
    assert "Enabling DuoAttention evaluation using sink size 16 and recent size 64" in out
    assert float(res["Average generation time"].split()[0]) > 0


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_reference_readme_quick_start_runs_as_printed(tmp_path):
    """The ```python block of the reference's README "Quick Start for DuoAttention" (README.md:119-153) is extracted and
    exec'd as printed — ``from duo_attn.utils import load_attn_pattern, sparsify_attention_heads``, ``from duo_attn.patch
    import enable_duo_attention_eval``, the keyword call ``sparsify_attention_heads(attn_heads, sparsity=0.5)``,
    ``enable_duo_attention_eval(model, attn_heads, sink_size=64, recent_size=256)`` — on the reference's own shipped
    Llama-3-8B-1048k pattern (copied into the temp directory at run time) and a random-init model of its geometry; then a
    43-token generation through the tuple caches: 128 of 256 kv heads retrieval heads, every layer's cache in the
    reference's format ``(full [2, nf, N, 128], streaming [2, ns, min(N, sink + recent), 128])`` (eviction has its own tests)"""
    res, out = _run("README", tmp_path)
    assert res == {"sparsity": "0.5", "retrieval heads": "128", "cache shapes ok": "True", "finite": "True", "tokens": "3"}
    assert "Enabling DuoAttention evaluation using sink size 64 and recent size 256" in out


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_reference_needle_in_a_haystack_runs_unchanged(tmp_path):
    """eval/needle/needle_in_haystack.py:176-330, 497-557 VERBATIM — the accuracy harness of the API north_star names:
    ``enable_duo_attention_eval(model, heads, sink, recent)`` with the command-line overrides, ``to_device(model, [gpu ids],
    enable_tp=True)`` on one device, chunked prefill handing ``output.past_key_values`` back and forth, the question fed one
    token at a time, greedy generation — two context lengths x two needle depths on a random-init model; the four result
    files it writes are read back.  (``rouge_score``, absent from the image, is a ten-line stand-in in the temp directory.)"""
    res, out = _run("needle", tmp_path)
    assert res == {"results": "4", "lengths": "[450, 600]", "depths": "[0.0, 100.0]", "fields ok": "True"}
    assert "Enabling DuoAttention evaluation using sink size 8 and recent size 24" in out
    assert out.count("-- Test Summary --") == 4


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
@pytest.mark.parametrize("method,file", [("duo_attn", "trec-duo_attn-pattern-pattern-sp-0.5.jsonl"), ("full", "trec-full.jsonl")])
def test_reference_longbench_pred_runs_unchanged(method, file, tmp_path):
    """eval/LongBench/pred.py:85-297 VERBATIM on task trec: ``--method duo_attn`` (load_attn_pattern, the keyword call
    ``sparsify_attention_heads(heads, None, sparsity=...)``, ``enable_duo_attention_eval`` with sink / recent overrides) and
    ``--method full`` (``duo_attn.patch.tuple_kv_cache.enable_tuple_kv_cache``: the full-attention tuple baseline, SURVEY row
    a12); ``to_device(model, [gpu ids], enable_tp=True)``; single-shot prefill, 7 prompt tokens fed one at a time, greedy
    generation; the prediction file it writes is read back (three records, the second truncated in the middle by the harness)."""
    res, out = _run(f"longbench:{method}", tmp_path)
    assert res == {"file": file, "records": "3", "fields ok": "True"}
    assert out.count("Prediction:") == 3
    assert ("Enabling DuoAttention evaluation using sink size 8 and recent size 24" in out) == (method == "duo_attn")


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_reference_needle_harness_with_two_devices_starts_its_own_ranks(tmp_path):
    """The reference starts its accuracy harnesses as ONE python process (scripts/niah.sh:17, scripts/longbench.sh) that hands
    every visible GPU to ``to_device(model, device_list, enable_tp=True)`` (eval/needle/needle_in_haystack.py:213-214).  Here that
    call — a device list longer than one, no process group, no rank environment — starts the SAME command line again as one
    rank per device under torch.distributed.run (duo_attn/launch.py: ensure_ranks), the ranks initialise their group inside
    to_device (gloo here: no GPU) and shard the already-patched model head-parallel, and rank 0 alone prints and writes: the
    four result files of two context lengths x two depths are read back, identical in form to the one-device run's."""
    res, out = _run("needle-tp2", tmp_path)
    assert res == {"results": "4", "lengths": "[450, 600]", "depths": "[0.0, 100.0]", "fields ok": "True"}, res
    assert out.count("RESULT ") == 1                     # rank 1 is silent
    import glob
    files = sorted(glob.glob(os.path.join(str(tmp_path), "results", "tiny-llama", "*_results.json")))
    assert len(files) == 4 and all(json.load(open(f))["model_response"] is not None for f in files)
