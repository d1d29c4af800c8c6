# round 3, GPU call 8: final build — whole GPU suite, smoke(), epilogue anatomy, bench line, kernel trace of the bench command
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3h
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
cp gpurun_out/parity_report.json $O/parity_r3.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_dtiming.so python tools/debug/decode_timing.py 4 0 2>&1 | grep -v amdgpu.ids | grep "prologue split\|epilogue split\|realtime\|per XCD: last" > $O/timing.txt; cat $O/timing.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -2 $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bt
rocprofv3 --kernel-trace --stats -d /tmp/bt -o b -- python $R/bench.py --steps 2 --warmup 1 --no-full-baseline --no-traffic --no-model-level --no-cpu-baseline --no-int4 > $O/bench_profiled.json 2> /tmp/bt.log
db=$(find /tmp/bt -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --top 10 > $O/kernels.md; else tail -5 /tmp/bt.log > $O/kernels.md; fi
cat $O/kernels.md
