#!/bin/bash
# Round 6, VERDICT item 2: parity of the new launch planner / block mapping, the launch map, and same-box A/B of the job at
# row blocks against the round-1..5 policy (DUO_PREFILL_PLANNER=0).   -> gpurun_out/r6_plan/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_plan; mkdir -p $O
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
timeout 1500 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py tests/test_batched_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.out; tail -5 $O/pytest.out
timeout 900 python tools/debug/prefill_launch_map.py --rows 1024 2048 4096 --json $O/map.json > $O/map.out 2> $O/map.err; tail -2 $O/map.out
timeout 600 python tools/debug/prefill_launch_map.py --rows 2048 --nf 1 3 4 6 --past 114688 --sweep > $O/sweep.out 2>> $O/map.err
for R in 0 4096 2048 1024; do
  for PL in 1 0; do
    RB=""; [ $R != 0 ] && RB="--row-block $R"
    DUO_PREFILL_PLANNER=$PL DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 $RB $LEAN > $O/job_R${R}_planner$PL.json 2>> $O/job.err
    python - <<PY
import json
d = json.load(open("$O/job_R${R}_planner$PL.json"))
print("R=$R planner=$PL", d["value"], d.get("prefill_tok_s") or d.get("config", {}), d["ms_per_step"])
PY
  done
done
# cfg3 at C = 4096, op level
for PL in 1 0; do
  DUO_PREFILL_PLANNER=$PL timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_planner$PL.json 2>> $O/job.err
done
tail -5 $O/job.err
