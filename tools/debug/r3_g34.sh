# round 3, GPU call 34: stdout of bench.py is the JSON line alone (host legs that print go to stderr)
cd $GRAFT_REPO_ROOT
timeout 150 python bench.py --steps 1 --warmup 0 --no-full-baseline --no-traffic --no-parity --no-int4 --no-kernel-roofline --no-model-level --no-token-linear --cpu-cfg1-layers 1 > /tmp/o.txt 2> /tmp/e.txt
echo "rc=$? stdout lines: $(wc -l < /tmp/o.txt)"; head -c 200 /tmp/o.txt; echo; grep -c "Enabling DuoAttention" /tmp/e.txt
python -c "import json; d=json.loads(open('/tmp/o.txt').read()); print(d['value'] > 0, d['cpu_baseline'].get('cfg1_end_to_end', {}).get('prefill_tok_s'))"
