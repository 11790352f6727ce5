#!/bin/bash
# round 4: with the border columns inside the interior launch, re-tune the interior grid: segmentation target (SBM_FAST_TARGET,
# workgroups per launch), two 128-disparity wavefronts at nd 256 (SBM_FAST_MODE=2), border chain length (SBM_DEV_BSEG, dev lib).
# usage: [SBM_LIB_AB=libsbm_hip_dev.so] tools/exp/r04_sweep.sh
one() {  # label workload
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
}
for wl in kitti ref640; do
  for t in 5600 8000 11000 15000 20000; do SBM_FAST_TARGET=$t one target=$t $wl; done
done
for wl in fhd uhd; do
  for m in 1 2; do
    for t in 5600 9000 14000; do SBM_FAST_MODE=$m SBM_FAST_TARGET=$t one mode=$m,target=$t $wl; done
  done
done
for wl in kitti ref640; do
  for b in 8 16 32 64 128; do SBM_DEV_BSEG=$b one bseg=$b $wl; done
done
