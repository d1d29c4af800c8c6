# round 3, GPU call 13: token-row linears of the decode step: parity, micro-benchmark vs torch, model-level decode A/B
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3m
mkdir -p $O
timeout 900 python -m pytest tests/test_token_linear_gpu.py -x -q 2>&1 | tail -15 > $O/pytest_linear.txt; cat $O/pytest_linear.txt
timeout 300 python tools/bench_kernels.py linear --reps 10 2>&1 | grep "^{" > $O/linear_bench.txt; cat $O/linear_bench.txt
timeout 300 python tools/bench_kernels.py linear --reps 10 --rows 4 2>&1 | grep "^{" > $O/linear_bench_rows4.txt; cat $O/linear_bench_rows4.txt
