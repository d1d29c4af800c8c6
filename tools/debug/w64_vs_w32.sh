#!/bin/bash
# w64 (debug bit 8: wherever legal) against the 8-wave kernel (bit 7) over launch shapes: tools/debug/w64_vs_w32.sh
for shape in "--nf 4 --past 0" "--nf 8 --past 0" "--nf 0 --past 65536" "--nf 4 --past 16384" "--nf 1 --past 114688" "--nf 4 --past 114688" "--nf 8 --past 114688"; do
  for f in 128 256; do
    echo -n "$shape flags=$f  "
    python tools/bench_kernels.py prefill $shape --chunk 16384 --reps 4 --flags $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.0f TF/s' % (d['avg_ms'], d['tflops_avg']))"
  done
done
