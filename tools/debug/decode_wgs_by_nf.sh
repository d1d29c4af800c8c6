#!/bin/bash
# decode scan: workgroups per launch x retrieval heads per layer (uniform layers), us per launch
for nf in 1 2 4 6 8; do
  for w in 128 192 256 384 512; do
    echo -n "nf=$nf wgs=$w  "
    DUO_DECODE_TARGET_WGS=$w python tools/bench_kernels.py decode --ctx 131072 --reps 20 --uniform-nf $nf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/launch  %.0f GB/s' % (d['avg_ms']*1000/8, d['GBps_avg']))"
  done
done
