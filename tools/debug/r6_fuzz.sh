#!/bin/bash
# round 6: drawn differential runs on the final tree (HIP vs oracle / module-by-module forms), fixed time budgets -> gpurun_out/r6_fuzz/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_fuzz; mkdir -p $O
run() { name=$1; shift; timeout 900 "$@" > $O/$name.out 2> $O/$name.err; echo "$name rc=$? $(tail -1 $O/$name.out | cut -c1-300)"; }
run static_a python tests/fuzz_static_path.py --seconds 300 --seed 6001
run static_big python tests/fuzz_static_path.py --seconds 200 --seed 6002 --big
run tuple python tests/fuzz_tuple_path.py --seconds 120 --seed 6003
run model_decode python tests/fuzz_model_decode.py --seconds 150 --seed 6004
run int4_decode python tests/fuzz_int4_decode.py --seconds 100 --seed 6005
run int4_cache python tests/fuzz_int4_cache.py --seconds 60 --seed 6006
run token_linear python tests/fuzz_token_linear.py --seconds 60 --seed 6007
