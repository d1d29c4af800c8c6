#!/bin/bash
# same-box A/B of prefill kernel builds: libs under duo-attention_amd/lib/ab/lib_<tag>.so, interleaved rounds
# usage: tools/debug/ab_prefill.sh [-n ROUNDS] tag1 tag2 ...
rounds=2
if [ "$1" = "-n" ]; then rounds=$2; shift 2; fi
for rep in $(seq $rounds); do
  for v in "$@"; do
    echo -n "$v  "
    DUO_ATTN_HIP_LIB=$PWD/duo-attention_amd/lib/ab/lib_$v.so python tools/bench_kernels.py prefill --nf 4 --past 65536 --chunk 16384 --reps 4 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.0f TF/s (best %.0f)' % (d['avg_ms'], d['tflops_avg'], d['tflops_best']))"
  done
done
