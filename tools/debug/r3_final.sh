# round 3: full GPU suite, smoke() and the default bench line on the final tree
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cp $R/gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python - <<'PY'
import json
d=[json.loads(l) for l in open('/root/repo/gpurun_out/r3final/bench.json') if l.startswith('{')][-1]
print({k:d.get(k) for k in ('value','ms_per_step','decode_ms_per_token')}, d['roofline'], d.get('roofline_decode'), d.get('model_level'))
PY
