# round 3, GPU call 29: packed-FMA arithmetic for 2-4 token rows: parity (all token-linear cases, model-level files), durations
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3ad
mkdir -p $O
timeout 1500 python -m pytest tests/test_token_linear_gpu.py tests/test_golden_and_model_gpu.py tests/test_sharded_models_gpu.py -x -q 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
for rows in 1 2 4; do
  rm -rf /tmp/prof_lin
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/bench_kernels.py linear --rows $rows --reps 6 > /tmp/lin_$rows.txt 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "token rows = $rows" >> $O/rows.md
  python $R/tools/rocpd_summary.py $db --top 20 | grep "token_linear" | cut -c1-150 >> $O/rows.md
done
cat $O/rows.md
