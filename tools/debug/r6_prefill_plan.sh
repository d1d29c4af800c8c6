#!/bin/bash
# Round 6, VERDICT item 2: parity of the new launch planner / block mapping, the launch map, and same-box A/B of the job at
# row blocks against the round-1..5 policy (DUO_PREFILL_PLANNER=0).   -> gpurun_out/r6_plan/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_plan; mkdir -p $O
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
timeout 1500 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py tests/test_batched_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.out; tail -5 $O/pytest.out
timeout 900 python tools/debug/prefill_launch_map.py --rows 1024 2048 4096 --json $O/map.json > $O/map.out 2> $O/map.err; tail -2 $O/map.out
timeout 600 python tools/debug/prefill_launch_map.py --rows 2048 --nf 1 3 4 6 --past 114688 --sweep > $O/sweep.out 2>> $O/map.err
# kernel durations of main and merge launches (device side)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_map && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_map -o p -- python $OLDPWD/tools/debug/prefill_launch_map.py --rows 2048 --nf 1 3 5 --past 114688 --reps 5 > /dev/null 2>> $OLDPWD/$O/map.err; python $OLDPWD/tools/rocpd_summary.py $(find /tmp/prof_map -name "*.db" | head -1) --top 8 > $OLDPWD/$O/map_kernels.md 2>> $OLDPWD/$O/map.err )
cat $O/map_kernels.md | head -20
for R in whole 4096 2048 1024; do
  for PL in 1 0; do
    if [ $R = whole ]; then
      DUO_PREFILL_PLANNER=$PL timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_R${R}_planner$PL.json 2>> $O/job.err
    else
      DUO_PREFILL_PLANNER=$PL DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R $LEAN > $O/job_R${R}_planner$PL.json 2>> $O/job.err
    fi
    python - <<PY
import json
d = json.load(open("$O/job_R${R}_planner$PL.json"))
print("R=$R planner=$PL", round(d["value"]), round(d.get("prefill_tok_s") or 0), round(d["ms_per_step"], 1))
PY
  done
done
# cfg3 at C = 4096, op level
for PL in 1 0; do
  DUO_PREFILL_PLANNER=$PL timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_planner$PL.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/cfg3_planner$PL.json')); print('cfg3 planner=$PL', round(d['value']), round(d.get('prefill_tok_s') or 0), d.get('roofline',{}).get('frac'), d.get('speedup_vs_full_attention') or d.get('vs_full'))"
done
grep -v amdgpu.ids $O/job.err | tail -5
