# round 3, GPU call 12: row sums of the bulk schedule as packed dots on the rounded P (W64_GEN_ROWSUM=dot2): parity, same-box A/B
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3l
mkdir -p $O
DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_dot2.so timeout 1200 python -m pytest tests/test_hip_kernels_gpu.py tests/test_golden_and_model_gpu.py tests/test_batched_gpu.py -x -q -k "prefill or static or golden or chunk" 2>&1 | tail -8 > $O/pytest_dot2.txt; cat $O/pytest_dot2.txt
bash tools/debug/ab_prefill.sh -n 3 base dot2 2>&1 | grep -v "^+" > $O/ab_dot2.txt; cat $O/ab_dot2.txt
