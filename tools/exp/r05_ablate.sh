#!/bin/bash
# round 5: marginal cost of three phases of the interior row loop (ablated builds lib/libsbm_hip_abl{1..4}.so: 1 = no neighbour
# selection tree, 2 = uniqueness deficits over a quarter of the registers, 3 = two exchange partners instead of NTERM - 1,
# 4 = all three; results of ablated builds are wrong by construction). usage (GPU box): bash tools/exp/r05_ablate.sh
for r in 1 2; do
for wl in ${WLS:-kitti ref640}; do
  for n in 0 1 2 3 4; do
    lib=libsbm_hip_abl$n.so; [ $n = 0 ] && lib=libsbm_hip.so
    SBM_LIB_AB=$lib python3 bench.py --no-cpu-baseline --workload $wl --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms']
print('abl$n', '$wl', 'ms/step', j['ms_per_step'], 'sad', s['sad'], j['roofline'].get('kernel'))"
  done
done
done
