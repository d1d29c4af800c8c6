# round 3, GPU call 2: prefill XCD-aware order (debug bit 10 = plain order) wall A/B + FETCH_SIZE + clock; decode preload A/B; decode anatomy
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3b
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels_gpu.py tests/test_full_size_gpu.py -x -q -k "prefill or row_block or static_hot or full_size or cfg" 2>&1 | tail -8 > $O/pytest_prefill.txt
cat $O/pytest_prefill.txt
# wall-time A/B per retrieval-head count (one layer-launch each, chunk 16384 at past 65536)
for rep in 1 2; do for nf in 3 4 5 6 2 8; do for f in 0 1024; do
  echo -n "nf=$nf flags=$f  "; python tools/bench_kernels.py prefill --nf $nf --past 65536 --chunk 16384 --reps 4 --flags $f 2>/dev/null | tail -1
done; done; done > $O/ab_xmap.txt 2>&1
cat $O/ab_xmap.txt
# whole job both ways
for rep in 1 2; do for f in 0 1024; do echo -n "DUO_DEBUG_FLAGS=$f  "; DUO_DEBUG_FLAGS=$f python bench.py --steps 2 --warmup 1 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-kernel-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','prefill_tok_s','decode_ms_per_token')})"; done; done > $O/ab_xmap_bench.txt 2>&1
cat $O/ab_xmap_bench.txt
# FETCH_SIZE + GRBM_GUI_ACTIVE per launch, both orders (own PMC passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
for nf in 3 4 6; do for f in 0 1024; do
  rm -rf /tmp/px_${nf}_$f
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/px_${nf}_$f -o p -- python $R/tools/bench_kernels.py prefill --nf $nf --past 65536 --chunk 16384 --reps 3 --flags $f > /tmp/px.log 2>&1
  db=$(find /tmp/px_${nf}_$f -name "*.db" | head -1)
  echo "## nf=$nf flags=$f" >> $O/xmap_pmc.md
  if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --pmc --top 2 >> $O/xmap_pmc.md; else tail -5 /tmp/px.log >> $O/xmap_pmc.md; fi
done; done
cat $O/xmap_pmc.md
cd $R
# decode: kernel-argument preload on / off (side library built without the flag), whole step
for rep in 1 2; do for lib in libduoattn_hip.so ab/lib_nopre.so; do echo -n "$lib  "; DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/$lib python bench.py --steps 2 --warmup 1 --no-full-baseline --no-cpu-baseline --no-traffic --no-model-level --no-parity --no-kernel-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','prefill_tok_s','decode_ms_per_token')})"; done; done > $O/ab_preload.txt 2>&1
cat $O/ab_preload.txt
for nf in 4 6; do for f in 0 512; do DUO_ATTN_HIP_LIB=$R/duo-attention_amd/lib/ab/lib_dtiming.so python tools/debug/decode_timing.py $nf $f; done; done > $O/timing.txt 2>&1
cat $O/timing.txt
# per-kernel durations of the decode step (scan + epilogue launch)
cd /tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- python $R/tools/debug/decode_step_loop.py > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/tools/rocpd_summary.py $db --top 6 > $O/decode_step_kernels.md; else tail -5 /tmp/kt.log > $O/decode_step_kernels.md; fi
cat $O/decode_step_kernels.md
