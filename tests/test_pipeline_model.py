"""tools/pipeline_model.py (the MODEL of bench.py --gpus N behind profiles/r6_scaling_model.md) runs from the committed one-GPU
bench line, reproduces that line at one GPU, and behaves like a pipeline: prefill speeds up with the stage count but by less
than the count, batch-1 decode gets slower by the hops, every stage's busy share is at most one; with two layer blocks per
rank (--virtual-stages 2) the prefill is never slower than with one and every layer is still owned exactly once."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipeline_model_is_consistent_with_its_inputs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pipeline_model.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    line = json.load(open(os.path.join(ROOT, "profiles", "r6_a_bench.json")))
    rows = {x["n_gpus"]: x for x in d["rows"] if "virtual_stages" not in x}
    two = {x["n_gpus"]: x for x in d["rows"] if x.get("virtual_stages") == 2}
    assert sorted(rows) == [1, 2, 4, 8] and sorted(two) == [2, 4, 8]
    one = rows[1]
    assert abs(one["prefill_tok_s"] / line["prefill_tok_s"] - 1) < 1e-6
    assert abs(one["decode_ms_per_token"] / line["decode_ms_per_token"] - 1) < 1e-6
    assert abs(one["job_tok_s"] / line["value"] - 1) < 2e-2
    for a, b in ((1, 2), (2, 4), (4, 8)):
        sp = rows[b]["prefill_tok_s"] / rows[a]["prefill_tok_s"]
        assert 1.3 < sp < 2.0, (a, b, sp)
        assert rows[b]["decode_ms_per_token"] > rows[a]["decode_ms_per_token"]
    for n, x in rows.items():
        assert len(x["stages"]) == n and x["stages"][0][0] == 0 and x["stages"][-1][1] == 32
        assert 0 < x["stage_busy_min_max"][0] <= x["stage_busy_min_max"][1] <= 1.0 + 1e-9
    for n, x in two.items():
        st = x["stages"]
        assert len(st) == 2 * n and st[0][0] == 0 and st[-1][1] == 32 and all(a[1] == b[0] for a, b in zip(st, st[1:]))
        assert x["group_size"] in (n, 2 * n)
        assert x["prefill_tok_s"] >= rows[n]["prefill_tok_s"] * 0.999          # the planner never picks a worse plan than one block
        assert x["decode_ms_per_token"] > rows[n]["decode_ms_per_token"]       # one more trip round the ring per token
        assert 0 < x["stage_busy_min_max"][0] <= x["stage_busy_min_max"][1] <= 1.0 + 1e-9
