#!/bin/bash
# round 4: LR kernel, iterations x block width at exact covers (nit = ceil(groups / bs)); libsbm_hip_devpost.so
export SBM_LIB_AB=libsbm_hip_devpost.so
one() {
  python3 bench.py --check --cpu-sample 8 --workload $2 --steps 100 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'lrcheck', s['lrcheck'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'))"
}
for r in 1 2; do
for cfg in "kitti 4 64" "kitti 4 128" "kitti 4 192" "kitti 2 128" "kitti 2 192" "kitti 2 256" "ref640 4 64" "ref640 4 128" "ref640 4 192" "ref640 2 128" "ref640 2 192" "fhd 4 128" "fhd 4 192" "fhd 4 256" "fhd 2 256" "uhd 4 256" "uhd 4 192" "uhd 4 320"; do
  set -- $cfg
  SBM_DEV_LR_PX=$2 SBM_DEV_LR_BS=$3 one px$2,bs$3 $1
done; done
