# round 3, GPU call 6: SURVEY 8(d) measurement list — cfg3 with both readings of the shipped Mistral pattern (op + model level),
# cfg1 end to end on the host, round-3 model-level kernel trace
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3f
mkdir -p $O
for pat in mistral-7b-v0.2@0.5 mistral-7b-v0.2@raw; do for C in 4096 32000; do
  python bench.py --pattern $pat --ctx 32768 --chunk $C --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-int4 2>/dev/null | tail -1 > $O/cfg3_op_${pat//@/_}_C$C.json
  python -c "import json,sys; d=json.load(open('$O/cfg3_op_${pat//@/_}_C$C.json')); print('$pat C=$C', {k:d[k] for k in ('value','prefill_tok_s','decode_ms_per_token','speedup_vs_full_attention')}, 'frac', d['roofline']['frac'], d['roofline_decode']['frac'])"
done; done 2>&1 | tee $O/cfg3_op.txt
for pat in mistral-7b-v0.2@0.5 mistral-7b-v0.2@raw; do
  python tools/benchmark_static.py --shape mistral-7b-v0.2 --max_length 32768 --prefilling_chunk_size 4096 --pattern $pat --graph --prefill_steps 2 --decode_steps 50 --decode_warmup 10 2>&1 | tail -4
done > $O/cfg3_model.txt 2>&1
python tools/benchmark_static.py --shape mistral-7b-v0.2 --max_length 32768 --prefilling_chunk_size 4096 --sparsity 0 --graph --prefill_steps 2 --decode_steps 50 --decode_warmup 10 2>&1 | tail -4 >> $O/cfg3_model.txt
cat $O/cfg3_model.txt
# model-level kernel trace of the 128K job (one prefill pass + decode through the graph)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mt
rocprofv3 --kernel-trace --stats -d /tmp/mt -o m -- python $R/tools/benchmark_static.py --max_length 131072 --prefilling_chunk_size 16384 --prefill_steps 1 --prefill_warmup 0 --decode_steps 20 --decode_warmup 5 > /tmp/mt.log 2>&1
db=$(find /tmp/mt -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --top 25 > $O/model_level_kernels.md; else tail -5 /tmp/mt.log > $O/model_level_kernels.md; fi
tail -4 /tmp/mt.log >> $O/model_level_kernels.md
cat $O/model_level_kernels.md
cd $R
# cfg1 end to end on the host (32 layers) — CPU only
python - <<'PY' > $O/cfg1_host_e2e.json 2> $O/cfg1_host_e2e.err
import json, sys
sys.argv = ["bench.py"]
import bench
print(json.dumps(bench.cpu_cfg1_end_to_end(32)))
PY
cat $O/cfg1_host_e2e.json; tail -3 $O/cfg1_host_e2e.err
