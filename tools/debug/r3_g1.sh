set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_hip_kernels_gpu.py -x -q -k "decode or fused or graph or static_hot or single_launch or state" 2>&1 | tail -15 > gpurun_out/r3a/pytest_decode.txt
cat gpurun_out/r3a/pytest_decode.txt
for rep in 1 2; do for f in 0 512; do echo "flags=$f"; python tools/bench_kernels.py decode --ctx 131072 --reps 20 --flags $f 2>/dev/null | tail -1; done; done > gpurun_out/r3a/ab_scan.txt 2>&1
cat gpurun_out/r3a/ab_scan.txt
for rep in 1 2; do for f in 0 512; do echo "DUO_DEBUG_FLAGS=$f"; DUO_DEBUG_FLAGS=$f python bench.py --steps 2 --warmup 1 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-kernel-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','prefill_tok_s','decode_ms_per_token')})"; done; done > gpurun_out/r3a/ab_bench.txt 2>&1
cat gpurun_out/r3a/ab_bench.txt
for nf in 4 1; do for f in 0 512; do DUO_ATTN_HIP_LIB=$PWD/duo-attention_amd/lib/ab/lib_dtiming.so python tools/debug/decode_timing.py $nf $f; done; done > gpurun_out/r3a/timing.txt 2>&1
cat gpurun_out/r3a/timing.txt
