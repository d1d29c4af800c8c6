#!/bin/bash
# same-box A/B of decode builds: libs under duo-attention_amd/lib/ab/lib_<tag>.so ("cur" = the tree's library)
rounds=2
if [ "$1" = "-n" ]; then rounds=$2; shift 2; fi
for rep in $(seq $rounds); do
  for v in "$@"; do
    lib=$PWD/duo-attention_amd/lib/ab/lib_$v.so; [ "$v" = cur ] && lib=$PWD/duo-attention_amd/lib/libduoattn_hip.so
    echo -n "$v  "
    DUO_ATTN_HIP_LIB=$lib python tools/bench_kernels.py decode --ctx 131072 --reps 20 2>/dev/null | tail -1 | cut -c1-200
  done
done
