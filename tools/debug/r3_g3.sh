# round 3, GPU call 3: whole GPU suite (incl. the new sharded-model tests), decode scan A/B (rotation, legacy kernel), anatomy
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3c
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
for rep in 1 2 3; do for f in 0 2048 512; do echo -n "DUO_DEBUG_FLAGS=$f  "; DUO_DEBUG_FLAGS=$f python bench.py --steps 2 --warmup 1 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-kernel-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','prefill_tok_s','decode_ms_per_token')})"; done; done > $O/ab_decode.txt 2>&1
cat $O/ab_decode.txt
for rep in 1 2; do for f in 0 2048 512; do echo -n "flags=$f  "; python tools/bench_kernels.py decode --ctx 131072 --reps 20 --flags $f 2>/dev/null | tail -1; done; done > $O/ab_scan.txt 2>&1
cat $O/ab_scan.txt
for nf in 4 6; do for f in 0 2048; do DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_dtiming.so python tools/debug/decode_timing.py $nf $f; done; done 2>&1 | grep -v amdgpu.ids > $O/timing.txt
cat $O/timing.txt
