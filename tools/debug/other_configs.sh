#!/bin/bash
# The rest of SURVEY §8(d)'s measurement list on the current build — what profiles/rN_other_configs.md is written from:
#   cfg3 (Mistral-7B-v0.2 shape, 32K context, both readings of "shipped attn_pattern") at op and model level,
#   cfg4's workload (the 1M-token job) on one GPU, and the full bench line with its kernel trace.
#   tools/debug/other_configs.sh <tag>      -> gpurun_out/<tag>_other/*.json, *.md
tag=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/${tag}_other
mkdir -p $O
cd $R
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity"
for p in mistral-7b-v0.2@0.5 mistral-7b-v0.2@raw; do
  for c in 4096 32000; do
    python bench.py --pattern $p --ctx 32768 --chunk $c --steps 3 --warmup 1 $LEAN > $O/cfg3_op_${p//[@.]/_}_c$c.json 2>> $O/err.log
  done
done
for a in "--pattern mistral-7b-v0.2@0.5" "--pattern mistral-7b-v0.2@raw" "--sparsity 0"; do
  n=$(echo $a | tr -c 'a-z0-9' '_')
  python tools/benchmark_static.py --shape mistral-7b-v0.2 --max_length 32768 --prefilling_chunk_size 4096 $a --graph \
      --all_decode_modes --prefill_steps 2 --prefill_warmup 1 --decode_steps 50 --decode_warmup 10 2>> $O/err.log | tail -1 > $O/cfg3_model_$n.json
done
# cfg4's workload on ONE GPU: 1 048 576-token chunked prefill (chunk 32 000, the reference's default) + 32 decode steps
python bench.py --ctx 1048576 --chunk 32000 --decode-tokens 32 --steps 1 --warmup 0 --no-full-baseline $LEAN > $O/cfg4_1m_single_gpu.json 2>> $O/err.log
# the full bench line of this build (what the driver runs) and the kernel trace of the same workload
# (the full bench line and its kernel trace: tools/debug/r6_final.sh)
tail -3 $O/err.log
ls -la $O
