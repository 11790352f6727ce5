#!/bin/bash
# round 6: baseline bench lines of a build (LIB=<file in lib/>, default the product): stage times of the bench workloads,
# small batches and the one-pair call. usage: LIB=libsbm_hip.so TAG=base bash tools/exp/r06_base.sh
O=gpurun_out/r06; mkdir -p $O; TAG=${TAG:-base}; LIB=${LIB:-libsbm_hip.so}
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$TAG', '$1', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'stages', {k: round(v,4) for k,v in s.items()}, 'check', j['cpu_baseline'].get('bit_exact_vs_gpu') if 'cpu_baseline' in j else None)"; }
for spec in "kitti 64" "ref640 64" "fhd 16" "uhd 4" "kitti 8" "kitti 1" "ref640 1"; do
  set -- $spec
  SBM_LIB_AB=$LIB python3 bench.py --check --cpu-sample 4 --workload $1 --pairs $2 --steps ${STEPS:-60} --warmup 5 2>/dev/null | line "$1x$2"
done | tee -a $O/base_$TAG.txt
