#!/bin/bash
# round 6, last call: the driver's test command, smoke() and bench command on the committed tree -> gpurun_out/r6_last/
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r6_last; mkdir -p $O
rm -f gpurun_out/model_rel.log gpurun_out/parity_report.json
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.out; tail -3 $O/pytest.out
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null; cp gpurun_out/model_rel.log $O/model_rel.log 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_driver_cmd.json')); r=d['roofline']; rd=d['roofline_decode']
print(round(d['value']), round(d['prefill_tok_s']), round(r['frac'],4), r['traffic'], round(d['decode_ms_per_token'],4), round(rd['frac'],4), round(rd['whole_step']['frac'],4), round(d['roofline_int4']['whole_step_3p3M']['frac'],4))"
