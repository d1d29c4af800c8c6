#!/bin/bash
# prefill kernel with forced / automatic key-range splits, small and large chunks
for cfg in "1 4096" "2 4096" "5 4096" "4 16384" "1 16384"; do set -- $cfg
  for k in 1 0; do   # DUO_PREFILL_KSPLIT=1 forces no split, 0 = automatic
    echo -n "nf=$1 chunk=$2 ksplit=$([ $k = 1 ] && echo off || echo auto)  "
    DUO_PREFILL_KSPLIT=$k python tools/bench_kernels.py prefill --nf $1 --past 114688 --chunk $2 --reps 4 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.0f TF/s' % (d['avg_ms'], d['tflops_avg']))"
  done
done
