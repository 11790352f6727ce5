#!/bin/bash
# sweep of the interior kernel's workgroup target (row segmentation) after the move to 5 wavefronts per SIMD
for wl in kitti ref640 fhd; do
  for t in 2800 3600 4400 5000 5600 6400 7200 8400 10000 12800; do
    SBM_FAST_TARGET=$t python3 bench.py --no-cpu-baseline --workload $wl --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms']
print('target $t', '$wl', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'border', s['border'])"
  done
done
