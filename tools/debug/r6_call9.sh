#!/bin/bash
# round 6, GPU call 9: the prefill kernel's bulk run chained across the pool / chunk boundary — parity, anatomy, same-process-family A/B (debug bit 21 = separate runs)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c9; mkdir -p $O
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
OLDF=$((1<<21))
timeout 1800 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py tests/test_batched_gpu.py tests/test_golden_and_model_gpu.py tests/test_fuzz_gpu.py tests/test_int4.py tests/test_int4_model_gpu.py tests/test_tuple_path_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.out
bash tools/debug/build_variant.sh wgtime -DW64_WGTIME > /dev/null 2>&1
for cfg in "2048 4 114688 16384" "16384 8 114688 16384" "4096 3 16384 4096"; do set -- $cfg
  for fl in 0 $OLDF; do echo "-- rows $1 nf $2 past $3 r1 $4 flags $fl"; DUO_ATTN_HIP_LIB=$PWD/duo-attention_amd/lib/ab/lib_wgtime.so timeout 300 python tools/debug/w64_wgtime.py --rows $1 --nf $2 --past $3 --r1 $4 --flags $fl 2>&1 | grep -v amdgpu.ids; done
done | tee $O/wgtime.txt
for rep in 1 2; do for R in whole 4096 2048 1024; do for FL in 0 $OLDF; do
  if [ $R = whole ]; then DUO_DEBUG_FLAGS=$FL timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_R${R}_f$FL.$rep.json 2>> $O/job.err
  else DUO_DEBUG_FLAGS=$FL DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R $LEAN > $O/job_R${R}_f$FL.$rep.json 2>> $O/job.err; fi
  python -c "
import json; d=json.load(open('$O/job_R${R}_f$FL.$rep.json')); print('R=$R flags=$FL rep=$rep', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))" | tee -a $O/jobs.txt
done; done; done
for FL in 0 $OLDF 0 $OLDF; do
  DUO_DEBUG_FLAGS=$FL timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_f$FL.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/cfg3_f$FL.json')); print('cfg3 flags=$FL', round(d['value']), round(d['prefill_tok_s']), round(d['roofline']['frac'],4), d['speedup_vs_full_attention']['prefill'])" | tee -a $O/jobs.txt
done
timeout 900 python tools/debug/prefill_launch_map.py --rows 1024 2048 4096 --json $O/map.json > $O/map.out 2> $O/map.err; tail -1 $O/map.out
grep -v amdgpu.ids $O/job.err | tail -3
