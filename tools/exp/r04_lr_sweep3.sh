#!/bin/bash
# round 4: LR kernel, the full-cover candidates the second sweep left out; libsbm_hip_devpost.so
export SBM_LIB_AB=libsbm_hip_devpost.so
one() {
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 100 --warmup 5 $3 $4 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'lrcheck', s['lrcheck'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'))"
}
for r in 1 2; do
for cfg in "ref640 2 64" "ref640 4 64" "fhd 2 192" "fhd 2 128" "fhd 4 256" "uhd 2 256" "uhd 4 256" "kitti 2 128" "kitti 4 256"; do
  set -- $cfg
  SBM_DEV_LR_PX=$2 SBM_DEV_LR_BS=$3 one px$2,bs$3 $1
done; done
