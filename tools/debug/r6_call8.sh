#!/bin/bash
# round 6, GPU call 8: decode split divisor at short contexts, GPU-side via the captured step
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_c8; mkdir -p $O
for div in 8 4 2; do
  echo "== DUO_DECODE_SPLIT_DIV=$div"
  DUO_DECODE_SPLIT_DIV=$div timeout 600 python tools/debug/decode_graph_sweep.py --pattern mistral-7b-v0.2@0.5 --ctx 1024 4096 8192 16384 32768 131072 2>&1 | grep -v amdgpu.ids
done | tee $O/decode_split_div.txt
