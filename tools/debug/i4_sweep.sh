#!/bin/bash
# sweep of the int4 decode kernel's tuning knobs (waves/SIMD, dequant mode) with and without the
# loads-only debug flag (bit 5)
for w in 2 3 4; do
  for mode in 0 1; do
    for fl in 0 32; do
      echo -n "W=$w mode=$mode flags=$fl  "
      DUO_INT4_DECODE_WAVES=$w DUO_INT4_DECODE_MODE=$mode python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 5 --flags $fl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.0f rows/us  %.0f GB/s' % (d['avg_ms'], d['rows_per_us'], d['GBps_avg']))"
    done
  done
done
