# round 3, GPU call 23: fused decode layers on tensor-parallel shards (2 ranks on cuda:0, gloo) + the other sharded-model tests
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3x
mkdir -p $O
timeout 1200 python -m pytest tests/test_sharded_models_gpu.py -x -q 2>&1 | tail -12 > $O/pytest_sharded.txt; cat $O/pytest_sharded.txt
