#!/bin/bash
# round 4: LR kernel variants (pixels per thread x widest block) on the four bench workloads; libsbm_hip_devpost.so
export SBM_LIB_AB=libsbm_hip_devpost.so
one() {
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 60 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'lrcheck', s['lrcheck'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'))"
}
for wl in kitti ref640 fhd uhd; do
  for px in 4 2; do for bs in 128 192 256 320; do
    SBM_DEV_LR_PX=$px SBM_DEV_LR_BS=$bs one px$px,bs$bs $wl
  done; done
done
