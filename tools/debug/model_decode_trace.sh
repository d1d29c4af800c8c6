#!/bin/bash
# Kernel traces of the whole-model decode step at 128K (random-init Llama-3-8B shape): the static path's reference loop (eager
# and auto-captured graph) and the tuple path (enable_duo_attention_eval) -> gpurun_out/<tag>_model_decode_{static,tuple}.md
tag=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for which in static tuple; do
  rm -rf /tmp/mdt_$which
  rocprofv3 --kernel-trace --stats -d /tmp/mdt_$which -o p -- python $R/tools/debug/decode_host_profile.py $which 131072 30 > $R/gpurun_out/${tag}_model_decode_$which.txt 2> /tmp/mdt_$which.err
  python $R/tools/rocpd_summary.py $(find /tmp/mdt_$which -name "*.db" | head -1) --top 16 > $R/gpurun_out/${tag}_model_decode_$which.md
  grep "ms/token" $R/gpurun_out/${tag}_model_decode_$which.txt
done
