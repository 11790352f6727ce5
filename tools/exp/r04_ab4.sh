#!/bin/bash
# round 4: A/B of two engine builds over the four bench workloads (A = libsbm_hip.so, B = $1), interleaved, 3 rounds
B=${1:-libsbm_hip_dev.so}
one() {  # label workload
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 60 --warmup 5 $3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', '$3', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
}
for r in 1 2 3; do
  for wl in kitti ref640 fhd uhd; do
    one A $wl
    SBM_LIB_AB=$B one B $wl
  done
done
