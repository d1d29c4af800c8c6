# round 3, GPU call 7: merge-kernel rework A/B, whole GPU suite, N > 1 rehearsal, final bench line + kernel trace
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3g
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
cp gpurun_out/parity_report.json $O/parity_r3.json 2>/dev/null
# decode step with the reworked merge launch (per-kernel durations)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- python $R/tools/debug/decode_step_loop.py > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --top 4 > $O/decode_step_kernels.md; else tail -5 /tmp/kt.log > $O/decode_step_kernels.md; fi
cat $O/decode_step_kernels.md
cd $R
python tools/bench_kernels.py decode_int4 --ctx 1048576 --reps 10 2>/dev/null | tail -1 > $O/int4_decode.txt; cat $O/int4_decode.txt
# N > 1 code path of bench.py rehearsed on one GPU (gloo hand-off, every rank on cuda:0): small job
DUO_BENCH_DEBUG_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 0 --ctx 32768 --chunk 8192 --layers 8 --decode-tokens 8 --no-full-baseline 2>$O/n2.err | tail -1 > $O/n2_rehearsal.json
python -c "import json; d=json.load(open('$O/n2_rehearsal.json')); print({k:d[k] for k in ('value','n_gpus','prefill_tok_s','decode_ms_per_token')}, d['pipeline'])"; tail -2 $O/n2.err
# the bench line of this build, then the same command under the kernel trace
python bench.py --cpu-cfg1-layers 1 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -2 $O/bench.err
cd /tmp
rm -rf /tmp/bt
rocprofv3 --kernel-trace --stats -d /tmp/bt -o b -- python $R/bench.py --steps 2 --warmup 1 --no-traffic --no-model-level --no-cpu-baseline > $O/bench_profiled.json 2> /tmp/bt.log
db=$(find /tmp/bt -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --top 12 > $O/kernels.md; else tail -5 /tmp/bt.log > $O/kernels.md; fi
cat $O/kernels.md
