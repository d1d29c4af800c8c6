# round 3, GPU call 21: new token_linear cases (LDS > 64 KiB, lm_head-sized), smoke(), 2 streaming waves per CU
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3v
mkdir -p $O
timeout 900 python -m pytest tests/test_token_linear_gpu.py -x -q 2>&1 | tail -12 > $O/pytest_linear.txt; cat $O/pytest_linear.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
for cfg in; do
  set -- $cfg
  rm -rf /tmp/prof_lin
  DUO_LINEAR_THREADS=$1 DUO_LINEAR_WAVES_PER_CU=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/benchmark_static.py --max_length 16384 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /dev/null 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "threads per workgroup = $1, streaming waves per CU = $2 (0 = automatic)" >> $O/sweep.md
  python $R/tools/rocpd_summary.py $db --top 14 | grep "token_linear" >> $O/sweep.md
done
cat $O/sweep.md
