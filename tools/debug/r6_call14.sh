#!/bin/bash
# round 6, GPU call 14: is the INT4 decode's loads-only ceiling (5.75 TB/s) the lane -> address order of its 16-byte loads?
# Timing probe (-DDUO_I4_COALESCED_PROBE, wrong results, loads-only flag): consecutive lanes fetch consecutive pieces.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c14; mkdir -p $O
SRC=duo_int4 bash tools/debug/build_variant.sh i4co -DDUO_I4_COALESCED_PROBE > /dev/null 2>&1
L=$PWD/duo-attention_amd/lib/ab
for rep in 1 2 3; do for lib in default i4co; do for w in 3 2 4; do
  if [ $lib = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$L/lib_$lib.so; fi
  echo -n "lib=$lib W=$w flags=32  "; DUO_INT4_DECODE_WAVES=$w timeout 300 python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 8 --flags 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.0f rows/us  %.0f GB/s' % (d['avg_ms'], d['rows_per_us'], d['GBps_avg']))"
done; done; done | tee $O/int4_probe.txt
