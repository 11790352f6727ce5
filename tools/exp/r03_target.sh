#!/bin/bash
# sweep of the interior kernel's workgroup target (row segmentation) for the single-wavefront 128-disparity kernel
for wl in ${WLS:-kitti}; do
  for t in ${TARGETS:-3000 4200 5600 6200 7200 8400 9300 10500 12400 15000}; do
    SBM_FAST_TARGET=$t python3 bench.py --no-cpu-baseline --workload $wl --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms']
print('target $t', '$wl', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'border', s['border'])"
  done
done
