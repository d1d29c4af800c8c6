#!/bin/bash
# round 6, final tree: the driver's own test command and bench command, the kernel trace of the bench job, the prefill
# kernel's counter passes, the row-block / cfg3 A/B against the round-1..5 launch policy, the other configs of SURVEY 8(d)
#   -> gpurun_out/r6_final/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R || exit 1
O=$R/gpurun_out/${OUT:-r6_final}; mkdir -p $O
rm -f gpurun_out/model_rel.log gpurun_out/parity_report.json
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.out; tail -3 $O/pytest.out
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null; cp gpurun_out/model_rel.log $O/model_rel.log 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
LEAN="--no-cpu-baseline --no-traffic --no-model-level --no-int4 --no-token-linear --no-parity --no-full-baseline"
for R_ in whole 4096 2048 1024; do for PL in 1 0; do
  if [ $R_ = whole ]; then DUO_PREFILL_PLANNER=$PL timeout 600 python bench.py --steps 3 --warmup 1 $LEAN > $O/job_R${R_}_planner$PL.json 2>> $O/job.err
  else DUO_PREFILL_PLANNER=$PL DUO_BENCH_FORCE_BLOCKS=1 timeout 600 python bench.py --steps 3 --warmup 1 --row-block $R_ $LEAN > $O/job_R${R_}_planner$PL.json 2>> $O/job.err; fi
  python -c "
import json; d=json.load(open('$O/job_R${R_}_planner$PL.json')); print('R=$R_ planner=$PL', round(d['value']), round(d['prefill_tok_s']), round(d['ms_per_step'],1))" | tee -a $O/jobs.txt
done; done
timeout 900 python tools/debug/prefill_launch_map.py --rows 1024 2048 4096 --json $O/map.json > $O/map.out 2> $O/map.err; tail -1 $O/map.out
# kernel trace of the bench job + the bench line of the same process
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r6 && rocprofv3 --kernel-trace --stats -d /tmp/prof_r6 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity > $O/prof_bench.json 2> /tmp/prof_r6.err; python $R/tools/rocpd_summary.py $(find /tmp/prof_r6 -name "*.db" | head -1) --top 14 > $O/kernels.md 2>> $O/job.err )
head -12 $O/kernels.md
# counter passes of the prefill kernel (separate passes, --pmc with --kernel-trace only)
( cd /tmp && export TMPDIR=/tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -o p -- python $R/tools/bench_kernels.py prefill --nf 4 --past 65536 --chunk 16384 --reps 3 > /tmp/pmc_$i.log 2>&1
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --pmc --top 3 > $O/pmc_$i.md; else tail -5 /tmp/pmc_$i.log > $O/pmc_$i.md; fi
done )
bash tools/debug/other_configs.sh ${OUT:-r6} > $O/other_configs.log 2>&1; tail -3 $O/other_configs.log
ls $O gpurun_out/${OUT:-r6}_other 2>/dev/null | head -60
