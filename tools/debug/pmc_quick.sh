#!/bin/bash
# quick PMC comparison of prefill builds: tools/debug/pmc_quick.sh <tag> ... (libs duo-attention_amd/lib/ab/lib_<tag>.so, W64 on)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "$@"; do rm -rf /tmp/q_$tag; DUO_PREFILL_W64=1 DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_$tag.so rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/q_$tag -o p -- python $R/tools/bench_kernels.py prefill --nf 4 --past 65536 --chunk 16384 --reps 2 > /tmp/q_$tag.log 2>&1; db=$(find /tmp/q_$tag -name "*.db" | head -1); echo "== $tag"; python $R/tools/rocpd_summary.py $db --pmc --top 2 | grep "prefill" | cut -c1-120; done
