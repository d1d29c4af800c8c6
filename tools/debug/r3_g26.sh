# round 3, GPU call 26: token-row linears, 8 loads per group (two groups = 16 KiB in flight per wave) against 4
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3aa
mkdir -p $O
DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_ling8.so timeout 600 python -m pytest tests/test_token_linear_gpu.py -x -q -k "case0 or case1 or case2 or case3 or case10 or case13 or rmsnorm_kernel" 2>&1 | tail -4 | tee $O/pytest_ling8.txt
cd /tmp && export TMPDIR=/tmp
for v in ling4 ling8 ling4 ling8; do
  rm -rf /tmp/prof_lin
  DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_$v.so timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/benchmark_static.py --max_length 16384 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /dev/null 2> /tmp/prof_lin.err
  db=$(find /tmp/prof_lin -name "*.db" | head -1)
  echo "$v" >> $O/sweep.md
  python $R/tools/rocpd_summary.py $db --top 14 | grep "token_linear" >> $O/sweep.md
done
cat $O/sweep.md
