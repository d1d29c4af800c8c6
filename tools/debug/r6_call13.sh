#!/bin/bash
# round 6, GPU call 13: INT4 decode with the four waves of a workgroup walking its chunk TILE-INTERLEAVED (-DDUO_I4_INTERLEAVE: one
# contiguous stream per workgroup and pool instead of four), alone and with three tile buffers: parity + same-box A/B  -> gpurun_out/r6_c13/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c13; mkdir -p $O
SRC=duo_int4 bash tools/debug/build_variant.sh i4il -DDUO_I4_INTERLEAVE > /dev/null 2>&1
SRC=duo_int4 bash tools/debug/build_variant.sh i4il3 -DDUO_I4_INTERLEAVE -DDUO_I4_DEPTH3 > /dev/null 2>&1
L=$PWD/duo-attention_amd/lib/ab
for v in i4il; do
DUO_ATTN_HIP_LIB=$L/lib_$v.so timeout 900 python -m pytest tests/test_int4.py tests/test_int4_golden.py tests/test_int4_model_gpu.py -x -q -m gpu -p no:cacheprovider > $O/int4_${v}_pytest.out 2>&1; echo "pytest($v) rc=$?"; tail -2 $O/int4_${v}_pytest.out
done
for rep in 1 2; do for lib in default i4il i4il3; do for fl in 0 32; do
  if [ $lib = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$L/lib_$lib.so; fi
  echo -n "lib=$lib flags=$fl  "; timeout 300 python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 8 --flags $fl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.0f rows/us  %.0f GB/s' % (d['avg_ms'], d['rows_per_us'], d['GBps_avg']))"
done; done; done | tee $O/int4_kernel.txt
for w in 2 4; do for lib in i4il; do export DUO_ATTN_HIP_LIB=$L/lib_$lib.so; for fl in 0 32; do
  echo -n "W=$w lib=$lib flags=$fl  "; DUO_INT4_DECODE_WAVES=$w timeout 300 python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 8 --flags $fl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.0f rows/us  %.0f GB/s' % (d['avg_ms'], d['rows_per_us'], d['GBps_avg']))"
done; done; done | tee -a $O/int4_kernel.txt
for lib in default i4il i4il3 default i4il i4il3; do
  if [ $lib = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$L/lib_$lib.so; fi
  echo -n "step lib=$lib "; timeout 600 python tools/debug/int4_legs.py step 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['whole_step_3p3M']; print(json.dumps({k: d[k] for k in ('ms_per_token','frac')}))"; done | tee $O/int4_step.txt
unset DUO_ATTN_HIP_LIB
timeout 300 python tests/fuzz_token_linear.py --case "{'rows': 1, 'n_in': 3736, 'sizes': (1889,), 'bias': False, 'pro': 'norm', 'residual': False, 'pad': 8, 'scale': 0.5, 'seed': 505371262}" 2>&1 | tail -2 | tee $O/token_linear_case.txt
timeout 300 python tests/fuzz_token_linear.py --seconds 120 --seed 6107 2>&1 | tail -2 | tee -a $O/token_linear_case.txt
