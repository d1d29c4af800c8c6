# round 3, GPU call 11: dynamic-deal experiment (DUO_DECODE_DYNAMIC=1): correctness of the decode tests, same-box A/B, XCD end times
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3k
mkdir -p $O
DUO_DECODE_DYNAMIC=1 timeout 900 python -m pytest tests/test_hip_kernels_gpu.py -x -q -k "test_decode or fused or static_hot" 2>&1 | tail -6 > $O/pytest_dyn.txt; cat $O/pytest_dyn.txt
for rep in 1 2 3; do for d in 0 1; do echo -n "DUO_DECODE_DYNAMIC=$d  "; DUO_DECODE_DYNAMIC=$d python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-kernel-roofline --no-int4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('decode_ms_per_token',)}, d['full_attention']['decode_tok_s'], d['speedup_vs_full_attention']['decode'])"; done; done > $O/ab_dyn.txt 2>&1
grep -v "^+" $O/ab_dyn.txt
for d in 0 1; do DUO_DECODE_DYNAMIC=$d DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_dtiming.so python tools/debug/decode_timing.py 4 0 2>&1 | grep "realtime\|per XCD: last"; done > $O/timing_dyn.txt
cat $O/timing_dyn.txt
