#!/bin/bash
# round 6, GPU call 3: INT4 decode at four waves per SIMD (A/B + parity), workgroup timelines of the prefill kernel, decode
# at 32K (one launch vs two), the graph test file and a short soak after the harness / test fixes  -> gpurun_out/r6_c3/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c3; mkdir -p $O
echo "== INT4 parity at 4 waves"
DUO_INT4_DECODE_WAVES=4 timeout 900 python -m pytest tests/test_int4.py tests/test_int4_golden.py tests/test_int4_model_gpu.py -x -q -m gpu -p no:cacheprovider > $O/int4_w4_pytest.out 2>&1; echo "rc=$?"; tail -3 $O/int4_w4_pytest.out
timeout 600 python -m pytest tests/test_int4.py tests/test_int4_golden.py -x -q -m gpu -p no:cacheprovider > $O/int4_w3_pytest.out 2>&1; echo "rc=$?"; tail -2 $O/int4_w3_pytest.out
echo "== INT4 timing"
for rep in 1 2; do for w in 3 4; do for fl in 0 32; do
  echo -n "W=$w flags=$fl  "; DUO_INT4_DECODE_WAVES=$w timeout 300 python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 8 --flags $fl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.0f rows/us  %.0f GB/s' % (d['avg_ms'], d['rows_per_us'], d['GBps_avg']))"
done; done; done | tee $O/int4_kernel.txt
for w in 3 4 3 4; do echo -n "step W=$w "; DUO_INT4_DECODE_WAVES=$w timeout 600 python tools/debug/int4_legs.py step 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['whole_step_3p3M']; print(json.dumps({k: d[k] for k in d if not isinstance(d[k], dict)}))"; done | tee $O/int4_step.txt
echo "== workgroup timelines"
for cfg in "2048 4 114688 16384" "2048 3 114688 16384" "16384 8 114688 16384" "16384 4 114688 16384" "2048 4 16384 16384"; do set -- $cfg
  echo "-- rows $1 nf $2 past $3 r1 $4"; DUO_ATTN_HIP_LIB=$PWD/duo-attention_amd/lib/ab/lib_wgtime.so timeout 300 python tools/debug/w64_wgtime.py --rows $1 --nf $2 --past $3 --r1 $4 2>&1 | grep -v amdgpu.ids
done | tee $O/wgtime.txt
echo "== decode at short contexts: one launch vs two, split cap"
for ctx in 4096 16384 32768; do for ol in 0 1; do
  echo -n "ctx=$ctx one_launch=$ol  "; DUO_DECODE_ONE_LAUNCH=$ol timeout 300 python bench.py --pattern mistral-7b-v0.2@0.5 --ctx $ctx --chunk 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('decode_ms_per_token'), d.get('decode_tok_s'))"
done; done | tee $O/decode_short.txt
echo "== graph test file + soak"
for i in 1 2 3 4 5 6; do timeout 600 python -X faulthandler -m pytest tests/test_auto_graph_gpu.py -x -q -m gpu -p no:cacheprovider > $O/graph_t$i.out 2>&1; echo "run $i rc=$? $(tail -1 $O/graph_t$i.out)"; done | tee $O/graph_runs.txt
timeout 400 python tools/debug/graph_soak.py --seconds 240 > $O/soak.out 2> $O/soak.err; echo "soak rc=$?"; tail -2 $O/soak.out
