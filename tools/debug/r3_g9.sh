# round 3, GPU call 9: INT4 B = 2 cache test; chunk-size sweep of the 128K job on the round-3 build
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3i
mkdir -p $O
timeout 600 python -m pytest tests/test_int4.py -x -q -k "batch_rows or chunked_prefill" 2>&1 | tail -15 > $O/pytest_int4_b2.txt; cat $O/pytest_int4_b2.txt
bash tools/debug/chunk_sweep.sh 4096 8192 32000 2>&1 | grep -v "^+" > $O/chunk_sweep.txt; cat $O/chunk_sweep.txt
