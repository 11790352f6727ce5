#!/bin/bash
# round 6: quick stage times of the two 64-pair workloads (+ one pair), per-kernel trace of the first
O=gpurun_out/r06; mkdir -p $O
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$1', 'ms/step', j['ms_per_step'], 'speckle', round(s['speckle'],4), 'lr', round(s['lrcheck'],4))"; }
for spec in "kitti 64" "ref640 64" "kitti 1"; do
  set -- $spec
  SBM_LIB_AB=${LIB:-libsbm_hip.so} python3 bench.py --no-cpu-baseline --workload $1 --pairs $2 --steps 60 --warmup 5 2>/dev/null | line "$1x$2"
done | tee -a $O/q.txt
bash tools/exp/r06_trace.sh "q1 ${LIB:-libsbm_hip.so} kitti 64" "q2 ${LIB:-libsbm_hip.so} ref640 64" 2>&1 | grep -i "speckle\|==" | cut -c1-40,100-170
