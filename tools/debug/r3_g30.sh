# round 3, GPU call 30: rocprofv3 --kernel-trace --stats of the bench job on the final tree (attention kernels + token-linear leg)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fin
rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o p -- python $R/bench.py --steps 1 --warmup 0 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity > $R/gpurun_out/r3_final_prof_bench.json 2> /tmp/prof_fin.err
db=$(find /tmp/prof_fin -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db --top 14 > $R/gpurun_out/r3_final_kernels.md
cat $R/gpurun_out/r3_final_kernels.md | cut -c1-200
python - <<'PY'
import json
d=[json.loads(l) for l in open('/root/repo/gpurun_out/r3_final_prof_bench.json') if l.startswith('{')][-1]
print(d['roofline']['avg_launch_ms'], d['roofline_decode']['avg_launch_ms'], d['roofline_token_linear']['roofline']['avg_launch_ms'], d['roofline_int4']['roofline']['avg_launch_ms'])
PY
