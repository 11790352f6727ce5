#!/bin/bash
# round 5: the product build -- full GPU test-suite, bench lines of every workload, envelope, small launches, host entry points
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputests6.txt; cat $O/gputests6.txt
for wl in kitti ref640 fhd uhd; do python3 bench.py --check --workload $wl 2>/dev/null | tail -1 > $O/bench6_$wl.json; python3 -c "
import json; j=json.load(open('$O/bench6_$wl.json')); r=j['roofline']
print('$wl', j['ms_per_step'], j.get('ms_per_step_median'), r['stage_ms'], j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"; done
bash tools/exp/r05_envelope.sh > $O/envelope6.txt 2>&1; cat $O/envelope6.txt
for wl in ref640 kitti fhd; do
  python3 bench.py --workload $wl --pairs 1 --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=1', d['ms_per_step'], d['roofline']['stage_ms'], d['roofline'].get('kernel'))"
done > $O/small6.txt 2>&1; cat $O/small6.txt
python3 tools/bench_host.py 2>/dev/null | tail -1 > $O/host_entry6.json; cat $O/host_entry6.json
