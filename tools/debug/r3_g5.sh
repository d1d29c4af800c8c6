# round 3, GPU call 5: batched entry points + whole GPU suite
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3e
mkdir -p $O
timeout 600 python -m pytest tests/test_batched_gpu.py -x -q 2>&1 | tail -25 > $O/pytest_batched.txt
cat $O/pytest_batched.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
