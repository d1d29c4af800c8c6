#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench job (duo job only) -> gpurun_out/<tag>_kernels.md + the bench line of the same process
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 1 --warmup 0 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity > $R/gpurun_out/${tag}_prof_bench.json 2> /tmp/prof_$tag.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db --top 12 > $R/gpurun_out/${tag}_kernels.md
