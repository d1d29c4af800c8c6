# round 3, GPU call 25: bench line with the token-linear traffic probe
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3z
mkdir -p $O
( time timeout 900 python bench.py --steps 1 --warmup 0 --no-full-baseline --no-cpu-baseline --no-parity > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt
tail -3 $O/time.txt; tail -5 $O/bench.err
python - <<'PY'
import json
d=[json.loads(l) for l in open('/root/repo/gpurun_out/r3z/bench.json') if l.startswith('{')][-1]
print(json.dumps(d.get('roofline_token_linear'), indent=1)); print(d['roofline']['traffic'], d['roofline']['traffic_source'])
PY
