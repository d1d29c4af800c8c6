#!/bin/bash
# prefill tok/s of the 128K job by chunk size, duo vs full attention: tools/debug/chunk_sweep.sh [chunks...]
for c in ${@:-4096 8192 32000}; do
  python bench.py --chunk $c --steps 1 --warmup 1 --no-model-level --no-cpu-baseline --no-traffic --no-parity 2>/dev/null > /tmp/cs_$c.json
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
d = json.load(open(f"/tmp/cs_{c}.json"))
print(c, round(d["prefill_tok_s"]), round(d["full_attention"]["prefill_tok_s"]), round(d["speedup_vs_full_attention"]["prefill"], 3), round(d["roofline"]["frac"], 3))
PY
done
