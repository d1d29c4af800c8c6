# round 3, GPU call 4: XCD-weighted decode partition sweep; INT4 fused tests; new bench legs smoke
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3d
mkdir -p $O
timeout 900 python -m pytest tests/test_int4_golden.py tests/test_int4.py tests/test_int4_model_gpu.py tests/test_abi_and_api.py -x -q 2>&1 | tail -6 > $O/pytest_int4.txt
cat $O/pytest_int4.txt
for rep in 1 2; do for dw in 0 2 4 6; do echo -n "odd_dw=$dw  "; DUO_DECODE_ODD_XCD_DW=$dw python bench.py --steps 2 --warmup 1 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-kernel-roofline --no-int4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','prefill_tok_s','decode_ms_per_token')})"; done; done > $O/ab_odd.txt 2>&1
cat $O/ab_odd.txt
for dw in 0 4; do DUO_DECODE_ODD_XCD_DW=$dw DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_dtiming.so python tools/debug/decode_timing.py 4 0; done 2>&1 | grep "realtime\|per XCD: last" > $O/timing_odd.txt
cat $O/timing_odd.txt
# the bench line with every leg (no host e2e), and a Mistral raw-pattern run
python bench.py --steps 2 --warmup 1 --cpu-cfg1-layers 0 > $O/bench_full.json 2> $O/bench_full.err; tail -c 3000 $O/bench_full.json; tail -3 $O/bench_full.err
