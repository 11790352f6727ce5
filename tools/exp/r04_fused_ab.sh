#!/bin/bash
# NOTE: needs commit f8c40e8 (the last one that still has the SBM_BORDER_FUSED switch and the side-stream border kernel); on later trees both runs take the same path.
# round 4: the border columns inside the interior launch (SBM_BORDER_FUSED=1, default) against the round-3 border kernel on
# the side stream (=0), alternating, bit-exact check on, per workload.  usage: tools/exp/r04_fused_ab.sh [workloads...]
WLS=${@:-kitti ref640 fhd uhd}
one() {  # tag workload
  python3 bench.py --check --cpu-sample 16 --workload $2 --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'border', s['border'], 'lr', s['lrcheck'], 'speckle', s['speckle'], 'pf', s['prefilter'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
}
for wl in $WLS; do
  for rep in 1 2; do
    SBM_BORDER_FUSED=0 one side $wl
    SBM_BORDER_FUSED=1 one fused $wl
  done
done
