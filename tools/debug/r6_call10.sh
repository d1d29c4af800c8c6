#!/bin/bash
# round 6, GPU call 10: the one drawn model-decode case that tripped the greedy-token check (a tie), and a measurement variant
# of the prefill kernel that requests the first two K/V tiles BEFORE the Q rows (-DW64_DMA_FIRST): parity, anatomy, A/B.
# (The switch was taken out of the kernel source after this measurement — no gain, profiles/r6_prefill_plan.md — so the
#  variant legs of this script only document what was run.)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c10; mkdir -p $O
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
CASE="{'family': 'llama', 'Hkv': 2, 'group': 4, 'inter': 72, 'L': 3, 'heads': [[1.0, 1.0], [1.0, 1.0], [0.0, 0.0]], 'bias': False, 'B': 2, 'sink': 64, 'recent': 8, 'prompt': 127, 'steps': 5, 'evict': True, 'seed': 1715245614}"
timeout 300 python tests/fuzz_model_decode.py --case "$CASE" 2>&1 | grep -v "amdgpu.ids\|Enabling" | tail -5 | tee $O/case.txt
timeout 400 python tests/fuzz_model_decode.py --seconds 150 --seed 6004 2>&1 | grep -v "amdgpu.ids\|Enabling" | tail -4 | tee $O/model_decode.txt
bash tools/debug/build_variant.sh dmafirst -DW64_DMA_FIRST > /dev/null 2>&1
bash tools/debug/build_variant.sh wgtime -DW64_WGTIME > /dev/null 2>&1
bash tools/debug/build_variant.sh wgtime_dmafirst -DW64_WGTIME -DW64_DMA_FIRST > /dev/null 2>&1
L=$PWD/duo-attention_amd/lib/ab
DUO_ATTN_HIP_LIB=$L/lib_dmafirst.so timeout 1500 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py tests/test_batched_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_dmafirst.out 2>&1; echo "pytest(dmafirst) rc=$?"; tail -2 $O/pytest_dmafirst.out
for cfg in "2048 4 114688 16384" "4096 3 16384 4096"; do set -- $cfg
  for v in wgtime wgtime_dmafirst; do echo "-- rows $1 nf $2 past $3 r1 $4 lib $v"; DUO_ATTN_HIP_LIB=$L/lib_$v.so timeout 300 python tools/debug/w64_wgtime.py --rows $1 --nf $2 --past $3 --r1 $4 2>&1 | grep -v amdgpu.ids; done
done | tee $O/wgtime.txt
for rep in 1 2; do for R in whole 2048 1024; do for V in default dmafirst; do
  if [ $V = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$L/lib_$V.so; fi
  if [ $R = whole ]; then timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_R${R}_$V.$rep.json 2>> $O/job.err
  else DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R $LEAN > $O/job_R${R}_$V.$rep.json 2>> $O/job.err; fi
  python -c "
import json; d=json.load(open('$O/job_R${R}_$V.$rep.json')); print('R=$R lib=$V rep=$rep', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))" | tee -a $O/jobs.txt
done; done; done
for V in default dmafirst default dmafirst; do
  if [ $V = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$L/lib_$V.so; fi
  timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_$V.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/cfg3_$V.json')); print('cfg3 lib=$V', round(d['value']), round(d['prefill_tok_s']), round(d['roofline']['frac'],4), d['speedup_vs_full_attention']['prefill'])" | tee -a $O/jobs.txt
done
grep -v amdgpu.ids $O/job.err | tail -3
