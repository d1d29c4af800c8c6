#!/bin/bash
# round 6, GPU call 16: whole chunks on alternating HIP streams (bench.py --block-streams N without row blocks): bit-equality,
# then the whole job with 1 / 2 / 3 streams (alternating processes), and cfg3
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c16; mkdir -p $O
timeout 900 python -m pytest tests/test_block_streams_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.out
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
for rep in 1 2; do for NS in 1 2 3; do
  timeout 600 python bench.py --steps 3 --warmup 1 --block-streams $NS $LEAN > $O/job_whole_s$NS.$rep.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/job_whole_s$NS.$rep.json')); print('R=whole streams=$NS rep=$rep', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))" | tee -a $O/jobs.txt
done; done
for NS in 1 2 1 2; do
  timeout 600 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx 32768 --chunk 4096 --steps 3 --warmup 1 --block-streams $NS --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity > $O/cfg3_s$NS.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/cfg3_s$NS.json')); print('cfg3 streams=$NS', round(d['value']), round(d['prefill_tok_s']), d['speedup_vs_full_attention']['prefill'])" | tee -a $O/jobs.txt
done
grep -v amdgpu.ids $O/job.err | tail -5
