#!/bin/bash
# soak of the opt-in automatic decode graph + repeated runs of its test file, fault evidence kept.
#   tools/debug/graph_soak.sh [soak seconds] [test-file runs]
# Everything lands under gpurun_out/soak/ (stderr of every run: faulthandler backtraces on a fatal signal).
cd "$(dirname "$0")/../.." || exit 1
S=${1:-480}; N=${2:-40}
O=gpurun_out/soak; mkdir -p $O
ulimit -c 0
export PYTHONFAULTHANDLER=1
echo "== soak, plain ($S s)"
timeout $((S + 120)) python tools/debug/graph_soak.py --seconds $S > $O/soak_plain.out 2> $O/soak_plain.err; echo "rc=$?" | tee -a $O/soak_plain.out
tail -3 $O/soak_plain.out
echo "== soak, AMD_SERIALIZE_KERNEL=3 ($((S / 4)) s)"
AMD_SERIALIZE_KERNEL=3 timeout $((S / 4 + 120)) python tools/debug/graph_soak.py --seconds $((S / 4)) > $O/soak_serialize.out 2> $O/soak_serialize.err; echo "rc=$?" | tee -a $O/soak_serialize.out
tail -3 $O/soak_serialize.out
echo "== soak, AMD_LOG_LEVEL=3 (2 rounds, log tail kept)"
AMD_LOG_LEVEL=3 timeout 600 python tools/debug/graph_soak.py --rounds 2 --replays 50 --seconds 400 > $O/soak_log3.out 2> $O/soak_log3.full; echo "rc=$?" | tee -a $O/soak_log3.out
wc -l $O/soak_log3.full | tee -a $O/soak_log3.out; tail -400 $O/soak_log3.full > $O/soak_log3.err.tail; grep -ci "error\|fault\|abort" $O/soak_log3.full | tee -a $O/soak_log3.out; rm -f $O/soak_log3.full
tail -3 $O/soak_log3.out
echo "== tests/test_auto_graph_gpu.py x $N"
ok=0; bad=0
for i in $(seq 1 $N); do
  if timeout 600 python -X faulthandler -m pytest tests/test_auto_graph_gpu.py -x -q -m gpu -p no:cacheprovider > $O/t_$i.out 2> $O/t_$i.err; then
    ok=$((ok + 1)); rm -f $O/t_$i.out $O/t_$i.err
  else
    bad=$((bad + 1)); echo "run $i FAILED rc=$?"; tail -30 $O/t_$i.out; tail -60 $O/t_$i.err
  fi
done
echo "test file runs: $ok clean, $bad failed" | tee $O/test_runs.txt
