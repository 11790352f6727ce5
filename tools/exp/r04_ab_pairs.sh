#!/bin/bash
# round 4: A/B of two engine builds ($1, $2 inside lib/) at pair counts that are not multiples of 8 (the strips of the last,
# incomplete group of pairs: round-robin over the XCDs against one contiguous eighth per XCD)
A=${1:-libsbm_hip_devA.so}; B=${2:-libsbm_hip_dev.so}
one() {  # label workload [--pairs n]
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 50 --warmup 5 $3 $4 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', '$3 $4', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
}
for r in 1 2; do
  for w in "uhd" "uhd --pairs 1" "uhd --pairs 2" "fhd --pairs 2" "fhd --pairs 4" "fhd --pairs 12" "kitti --pairs 12" "kitti --pairs 4" "ref640 --pairs 1" "ref640 --pairs 4"; do
    SBM_LIB_AB=$A one A $w
    SBM_LIB_AB=$B one B $w
  done
done
