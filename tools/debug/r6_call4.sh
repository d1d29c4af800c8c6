#!/bin/bash
# round 6, GPU call 4: in-kernel merge of split prefill launches (parity, A/B against the launch pair), decode GPU-side sweep
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c4; mkdir -p $O
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
timeout 1800 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py tests/test_batched_gpu.py tests/test_golden_and_model_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.out
timeout 900 python tools/debug/prefill_launch_map.py --rows 1024 2048 4096 --json $O/map.json > $O/map.out 2> $O/map.err; tail -1 $O/map.out
timeout 600 python tools/debug/prefill_launch_map.py --rows 2048 --nf 1 3 4 6 --past 114688 --sweep > $O/sweep.out 2>> $O/map.err
for R in whole 4096 2048 1024; do for IK in 1 0; do
  if [ $R = whole ]; then DUO_PREFILL_INKERNEL_MERGE=$IK timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_R${R}_ik$IK.json 2>> $O/job.err
  else DUO_PREFILL_INKERNEL_MERGE=$IK DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R $LEAN > $O/job_R${R}_ik$IK.json 2>> $O/job.err; fi
  python -c "
import json; d=json.load(open('$O/job_R${R}_ik$IK.json')); print('R=$R inkernel=$IK', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))"
done; done
for IK in 1 0; do
  DUO_PREFILL_INKERNEL_MERGE=$IK timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_ik$IK.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/cfg3_ik$IK.json')); print('cfg3 inkernel=$IK', round(d['value']), round(d['prefill_tok_s']), d['roofline']['frac'], d['speedup_vs_full_attention'])"
done
echo "== decode sweep (graph vs eager, one vs two launches)"
timeout 900 python tools/debug/decode_graph_sweep.py --pattern mistral-7b-v0.2@0.5 2>&1 | grep -v amdgpu.ids | tee $O/decode_sweep_mistral.txt
timeout 900 python tools/debug/decode_graph_sweep.py --pattern llama3-8b-1048k@0.5 --ctx 32768 131072 2>&1 | grep -v amdgpu.ids | tee $O/decode_sweep_llama.txt
grep -v amdgpu.ids $O/job.err | tail -5
