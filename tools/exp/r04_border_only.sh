#!/bin/bash
# round 4: duration of the border wavefronts alone (development build lib/libsbm_hip_dev.so = -DSBM_DEV -DSBM_DEV_FEW of
# sbm_sad_fast.hip; SBM_DEV_BORDER_ONLY cuts the grid after the border workgroups, results are wrong by construction)
for wl in ${@:-kitti ref640 fhd uhd}; do
  SBM_LIB_AB=libsbm_hip_dev.so SBM_DEV_BORDER_ONLY=1 python3 bench.py --no-cpu-baseline --workload $wl --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms']
print('border-only', '$wl', 'sad stage', s['sad'], j['engine_library'])"
done
