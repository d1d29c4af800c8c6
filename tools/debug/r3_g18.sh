# round 3, GPU call 18: token-row linears: waves-per-CU sweep (device-side durations, eager decode steps at 32K), eager vs graph decode
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 0 8 12 16 24 32; do
  rm -rf /tmp/prof_lin
  DUO_LINEAR_WAVES_PER_CU=$w timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/benchmark_static.py --max_length 16384 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /dev/null 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "waves per CU = $w (0 = automatic)" >> $O/sweep.md
  python $R/tools/rocpd_summary.py $db --top 14 | grep "token_linear" >> $O/sweep.md
done
cat $O/sweep.md
cd $R
timeout 600 python tools/benchmark_static.py --max_length 32768 --prefill_steps 1 --prefill_warmup 0 --decode_steps 50 --decode_warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager 32K', d['avg_generation_time_ms'])" | tee $O/eager.txt
