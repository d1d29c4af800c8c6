# round 3, GPU call 28: token-row linears at 2 and 4 token rows (device-side durations)
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3ac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rows in 3 4; do
  rm -rf /tmp/prof_lin
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/bench_kernels.py linear --rows $rows --reps 6 > /tmp/lin_$rows.txt 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "token rows = $rows" >> $O/rows.md
  python $R/tools/rocpd_summary.py $db --top 20 | grep "token_linear\|Cijk\|elementwise\|rmsnorm" | cut -c1-150 >> $O/rows.md
done
cat $O/rows.md
