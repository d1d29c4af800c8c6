#!/bin/bash
# round 6, GPU call 11: fabric traffic of the prefill launch with the chained bulk run (default) and without (debug bit 21), same box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c11; mkdir -p $O
LEAN="--no-cpu-baseline --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
for rep in 1 2; do for FL in 0 2097152; do
  DUO_DEBUG_FLAGS=$FL timeout 900 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_f$FL.$rep.json 2>> $O/job.err
  python -c "
import json; d=json.load(open('$O/job_f$FL.$rep.json')); r=d['roofline']; print('flags=$FL rep=$rep', round(d['value']), round(d['prefill_tok_s']), round(r['frac'],4), 'traffic', r['traffic'], 'launch ms', round(r['avg_launch_ms'],3))" | tee -a $O/jobs.txt
done; done
grep -v amdgpu.ids $O/job.err | tail -3
