#!/bin/bash
# round 6, GPU call 12: INT4 decode with THREE tile buffers per wave (two tiles of fetch lead, -DDUO_I4_DEPTH3, 154 VGPRs at three
# waves per SIMD): parity + same-box A/B against the default (two buffers)  -> gpurun_out/r6_c12/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c12; mkdir -p $O
SRC=duo_int4 bash tools/debug/build_variant.sh i4d3 -DDUO_I4_DEPTH3 > /dev/null 2>&1
V=$PWD/duo-attention_amd/lib/ab/lib_i4d3.so
DUO_ATTN_HIP_LIB=$V timeout 900 python -m pytest tests/test_int4.py tests/test_int4_golden.py tests/test_int4_model_gpu.py -x -q -m gpu -p no:cacheprovider > $O/int4_d3_pytest.out 2>&1; echo "pytest(d3) rc=$?"; tail -2 $O/int4_d3_pytest.out
for rep in 1 2; do for lib in default d3; do for fl in 0 32; do
  if [ $lib = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$V; fi
  echo -n "lib=$lib flags=$fl  "; timeout 300 python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 8 --flags $fl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.0f rows/us  %.0f GB/s' % (d['avg_ms'], d['rows_per_us'], d['GBps_avg']))"
done; done; done | tee $O/int4_kernel.txt
for lib in default d3 default d3; do
  if [ $lib = default ]; then unset DUO_ATTN_HIP_LIB; else export DUO_ATTN_HIP_LIB=$V; fi
  echo -n "step lib=$lib "; timeout 600 python tools/debug/int4_legs.py step 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['whole_step_3p3M']; print(json.dumps({k: d[k] for k in d if not isinstance(d[k], dict)}))"; done | tee $O/int4_step.txt
