#!/bin/bash
# round 4, second sweep: finer segmentation targets, and what the border columns cost inside the launch (--no-postfilter: no LR
# check, hence no border work at all -- the pure interior kernel)
one() {  # label workload extra-args
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 40 --warmup 5 $3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
}
for wl in kitti ref640 fhd uhd; do
  one interior-only $wl --no-postfilter
  one with-border $wl
done
for t in 16000 24000 32000 48000; do SBM_FAST_TARGET=$t one target=$t kitti; SBM_FAST_TARGET=$t one target=$t,interior-only kitti --no-postfilter; done
for wl in fhd uhd; do for t in 2500 4000; do SBM_FAST_TARGET=$t one target=$t $wl; done; done
