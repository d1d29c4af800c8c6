# round 3, GPU call 20: token-row linears with 4 streaming waves per CU (staging by all waves of the workgroup): parity, durations, model
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3t
mkdir -p $O
timeout 900 python -m pytest tests/test_token_linear_gpu.py -x -q 2>&1 | tail -12 > $O/pytest_linear.txt; cat $O/pytest_linear.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "1024 8" "1024 16" "512 4"; do
  set -- $cfg
  rm -rf /tmp/prof_lin
  DUO_LINEAR_THREADS=$1 DUO_LINEAR_WAVES_PER_CU=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/benchmark_static.py --max_length 16384 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /dev/null 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "threads per workgroup = $1, streaming waves per CU = $2 (0 = automatic)" >> $O/sweep.md
  python $R/tools/rocpd_summary.py $db --top 14 | grep "token_linear" >> $O/sweep.md
done
cat $O/sweep.md
cd $R
timeout 900 python tools/benchmark_static.py --graph --prefill_steps 1 --prefill_warmup 0 --decode_steps 100 --decode_warmup 20 2>/dev/null | tail -1 > $O/model.json
python -c "import json; d=json.load(open('$O/model.json')); print({k: d[k] for k in ('avg_generation_time_ms',)})"
