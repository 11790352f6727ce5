#!/bin/bash
# round 5: multi-wavefront strips with ONE workgroup barrier per row (a row is finished behind the next row's winner-merge
# barrier; lib/libsbm_hip_lf1.so) against two barriers (lib/libsbm_hip_lf0.so): parity of the one-barrier build, then A/B
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3   # lib/libsbm_hip.so = the tree (one barrier)
LIBS="libsbm_hip_lf0.so libsbm_hip_lf1.so" WLS="fhd uhd" ROUNDS=3 bash tools/exp/r05_ab.sh
for r in 1 2; do for lib in libsbm_hip_lf0.so libsbm_hip_lf1.so; do
  for x in "--workload kitti --ndisp 256" "--workload fhd --ndisp 256" "--workload ref640 --ndisp 128 --block 21 --pairs 64"; do
  SBM_LIB_AB=$lib python3 bench.py --check --cpu-sample 4 --steps 30 --warmup 5 $x 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$lib', '$x', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
done; done; done
