# round 3, GPU call 16: token-row linears, revised prologue (token rows requested before the weight groups, epilogue operands first)
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3p
mkdir -p $O
timeout 900 python -m pytest tests/test_token_linear_gpu.py -x -q 2>&1 | tail -25 > $O/pytest_linear.txt; cat $O/pytest_linear.txt
DUO_FUSED_DECODE_LAYER=1 timeout 900 python tools/benchmark_static.py --graph --prefill_steps 1 --prefill_warmup 0 --decode_steps 100 --decode_warmup 20 2>/dev/null | tail -1 > $O/model_fused1.json
python -c "import json; d=json.load(open('$O/model_fused1.json')); print('fused=1', {k: d[k] for k in ('avg_generation_time_ms','avg_context_time_ms')})"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lin
DUO_FUSED_DECODE_LAYER=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o p -- python $R/tools/benchmark_static.py --max_length 32768 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /dev/null 2> /tmp/prof_lin.err
db=$(find /tmp/prof_lin -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db --top 14 | grep "token_linear\|kernel |\|---" > $O/model_decode_kernels_fused.md
cat $O/model_decode_kernels_fused.md
