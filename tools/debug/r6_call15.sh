#!/bin/bash
# round 6, GPU call 15: consecutive row blocks on alternating HIP streams (bench.py --block-streams N, one GPU, measurement):
# bit-equality test, then the whole job at 4096- / 2048- / 1024-row blocks with 1 / 2 / 3 streams, planner on and off
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c15; mkdir -p $O
timeout 900 python -m pytest tests/test_block_streams_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.out
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_whole.json 2>> $O/job.err
python -c "
import json; d=json.load(open('$O/job_whole.json')); print('R=whole', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))" | tee -a $O/jobs.txt
for R in 4096 2048 1024; do for NS in 1 2 3; do for PL in 1 0; do
  DUO_PREFILL_PLANNER=$PL DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R --block-streams $NS $LEAN > $O/job_R${R}_s${NS}_p$PL.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/job_R${R}_s${NS}_p$PL.json')); print('R=$R streams=$NS planner=$PL', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))" | tee -a $O/jobs.txt
done; done; done
for NS in 1 2; do DUO_DEBUG_FLAGS=256 DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block 2048 --block-streams $NS $LEAN > $O/job_R2048_s${NS}_nosplit.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/job_R2048_s${NS}_nosplit.json')); print('R=2048 streams=$NS no splits', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))" | tee -a $O/jobs.txt
done
grep -v amdgpu.ids $O/job.err | tail -5
