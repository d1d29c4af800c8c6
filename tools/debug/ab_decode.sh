#!/bin/bash
# same-box A/B of the bench job's decode step between two builds of the library:  tools/debug/ab_decode.sh <old.so> [rounds]
cd "$(dirname "$0")/../.."
OLD=$1; N=${2:-3}
for i in $(seq $N); do
  for lib in "$OLD" duo-attention_amd/lib/libduoattn_hip.so; do
    echo -n "$lib: "
    DUO_ATTN_HIP_LIB=$PWD/$lib python bench.py --steps 2 --warmup 1 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level \
        --no-int4 --no-token-linear --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('decode ms/token %.4f   whole step frac %s' % (d['decode_ms_per_token'], d['roofline_decode'].get('whole_step', {}).get('frac')))"
  done
done
