#!/bin/bash
# PMC passes over the prefill micro-benchmark (one layer-launch shape of the bench job), separate passes
# (no trace domains mixed with --pmc).  Output: gpurun_out/pmc_<tag>_<n>.md
tag=${1:-base}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lib=$R/duo-attention_amd/lib/ab/lib_$tag.so
[ -f "$lib" ] || lib=$R/duo-attention_amd/lib/libduoattn_hip.so
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  DUO_PREFILL_W64=${W64:-0} DUO_ATTN_HIP_LIB=$lib rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -o p -- python $R/tools/bench_kernels.py prefill --nf 4 --past 65536 --chunk 16384 --reps 3 > /tmp/pmc_$i.log 2>&1
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --pmc --top 3 > $R/gpurun_out/pmc_${tag}_$i.md; else tail -5 /tmp/pmc_$i.log > $R/gpurun_out/pmc_${tag}_$i.md; fi
done
