#!/bin/bash
# round 3 A/B helper: bench lines (ms per step, stage ms) of the current build against lib/libsbm_hip_base.so (a build of
# an earlier commit, selected with SBM_LIB_AB), alternating, with the oracle check.
# usage: tools/exp/r03_ab.sh TAG [workloads...]     (run on the GPU box from the repo root)
TAG=${1:-ab}; shift
WLS=${@:-kitti ref640}
one() {  # lib-tag workload
  python3 bench.py --check --cpu-sample 16 --workload $2 --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'border', s['border'], 'lr', s['lrcheck'], 'speckle', s['speckle'], 'pf', s['prefilter'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'))"
}
for wl in $WLS; do
  for rep in 1 2; do
    [ -f u96-slam_amd/lib/libsbm_hip_base.so ] && SBM_LIB_AB=libsbm_hip_base.so one base $wl
    one $TAG $wl
  done
done
