#!/bin/bash
# round 5: exact-count kernels for 384 / 512 disparities (lib/libsbm_hip_ex1.so, -DSBM_FAST_EXACT512=1) against the masked-count
# ones (ex0): every line checked against the oracle
for r in 1 2; do for x in "--workload fhd --ndisp 512" "--workload kitti --ndisp 512 --pairs 32" "--workload uhd --ndisp 512" "--workload fhd --ndisp 384" "--workload kitti --ndisp 384 --pairs 32"; do
  for lib in libsbm_hip_ex0.so libsbm_hip_ex1.so; do
  SBM_LIB_AB=$lib python3 bench.py --check --cpu-sample 1 --steps 20 --warmup 3 $x 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$lib', '$x', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
done; done; done
