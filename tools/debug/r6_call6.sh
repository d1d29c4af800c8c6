#!/bin/bash
# round 6, GPU call 6: general-tile LDS-DMA at the start of the tile — parity, anatomy, same-box A/B against the previous placement
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c6; mkdir -p $O
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
OLD=$PWD/duo-attention_amd/lib/ab/lib_olddma.so
timeout 1500 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py tests/test_batched_gpu.py tests/test_int4.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.out
for cfg in "2048 4 114688 16384" "16384 8 114688 16384" "16384 4 114688 16384" "4096 3 16384 4096"; do set -- $cfg
  echo "-- rows $1 nf $2 past $3 r1 $4"; DUO_ATTN_HIP_LIB=$PWD/duo-attention_amd/lib/ab/lib_wgtime.so timeout 300 python tools/debug/w64_wgtime.py --rows $1 --nf $2 --past $3 --r1 $4 2>&1 | grep -v amdgpu.ids
done | tee $O/wgtime.txt
timeout 900 python tools/debug/prefill_launch_map.py --rows 1024 2048 4096 --json $O/map.json > $O/map.out 2> $O/map.err; tail -1 $O/map.out
timeout 600 python tools/debug/prefill_launch_map.py --rows 2048 --nf 1 3 4 6 --past 114688 --sweep > $O/sweep.out 2>> $O/map.err
for rep in 1 2; do for R in whole 2048; do for LIB in new old; do
  E=""; [ $LIB = old ] && E="DUO_ATTN_HIP_LIB=$OLD"
  if [ $R = whole ]; then env $E timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_R${R}_$LIB$rep.json 2>> $O/job.err
  else env $E DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R $LEAN > $O/job_R${R}_$LIB$rep.json 2>> $O/job.err; fi
  python -c "
import json; d=json.load(open('$O/job_R${R}_$LIB$rep.json')); print('R=$R lib=$LIB rep=$rep', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))"
done; done; done
for R in 4096 1024; do DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R $LEAN > $O/job_R${R}_new.json 2>> $O/job.err; python -c "
import json; d=json.load(open('$O/job_R${R}_new.json')); print('R=$R lib=new', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))"; done
for LIB in new old; do E=""; [ $LIB = old ] && E="DUO_ATTN_HIP_LIB=$OLD"
  env $E timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_$LIB.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/cfg3_$LIB.json')); print('cfg3 lib=$LIB', round(d['value']), round(d['prefill_tok_s']), d['roofline']['frac'], d['decode_ms_per_token'], d['roofline_decode'].get('captured_step',{}).get('ms_per_token'), d['speedup_vs_full_attention'])"
done
grep -v amdgpu.ids $O/job.err | tail -5
