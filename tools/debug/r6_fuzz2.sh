#!/bin/bash
# round 6: second, longer drawn campaign on the final tree (new seeds) + the graph soak's short form -> gpurun_out/r6_fuzz2/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_fuzz2; mkdir -p $O
run() { name=$1; shift; timeout 1500 "$@" > $O/$name.out 2> $O/$name.err; echo "$name rc=$? $(tail -1 $O/$name.out | cut -c1-300)"; }
run static_a python tests/fuzz_static_path.py --seconds 420 --seed 6101
run static_big python tests/fuzz_static_path.py --seconds 300 --seed 6102 --big
run tuple python tests/fuzz_tuple_path.py --seconds 240 --seed 6103
run model_decode python tests/fuzz_model_decode.py --seconds 360 --seed 6104
run int4_decode python tests/fuzz_int4_decode.py --seconds 240 --seed 6105
run int4_cache python tests/fuzz_int4_cache.py --seconds 120 --seed 6106
run token_linear python tests/fuzz_token_linear.py --seconds 120 --seed 6107
