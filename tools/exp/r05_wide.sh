#!/bin/bash
# round 5: the sliding-sum fallback kernel (sbm_sad_wide.hip) against the per-column one (SBM_WIDE=0) outside the fast envelope:
# parity tests, then ms per step at the bench shape. usage (GPU box): bash tools/exp/r05_wide.sh
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_wide.py -x -q -m gpu 2>&1 | tail -5
for x in "--workload kitti --block 31 --pairs 16" "--workload kitti --ndisp 512 --pairs 16" "--workload ref640 --block 41 --pairs 16" "--workload fhd --ndisp 512 --block 31 --pairs 4"; do
  for wide in 1 0; do
    SBM_WIDE=$wide python3 bench.py --check --cpu-sample 1 --steps 5 --warmup 1 $x 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('SBM_WIDE=$wide', '$x', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'value', j['value'], j['unit'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
  done
done
