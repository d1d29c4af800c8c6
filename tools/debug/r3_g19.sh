# round 3, GPU call 19: token-row linears: waves-per-CU / workgroup-size sweep, second part
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "256 4" "256 8" "512 8" "512 16" "1024 16" "512 4"; do
  set -- $cfg
  rm -rf /tmp/prof_lin
  DUO_LINEAR_THREADS=$1 DUO_LINEAR_WAVES_PER_CU=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/benchmark_static.py --max_length 16384 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /dev/null 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "threads per workgroup = $1, waves per CU = $2" >> $O/sweep.md
  python $R/tools/rocpd_summary.py $db --top 14 | grep "token_linear" >> $O/sweep.md
done
cat $O/sweep.md
