#!/bin/bash
# round 5: A/B of engine builds inside lib/ over bench workloads, interleaved. usage: LIBS="a.so b.so" WLS="kitti ref640" ROUNDS=2 bash tools/exp/r05_ab.sh
LIBS=${LIBS:-"libsbm_hip.so libsbm_hip_dev.so"}
for r in $(seq 1 ${ROUNDS:-2}); do
  for wl in ${WLS:-kitti ref640 fhd uhd}; do
    for lib in $LIBS; do
      SBM_LIB_AB=$lib python3 bench.py --check --cpu-sample 8 --workload $wl --steps 60 --warmup 5 $BENCH_EXTRA 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$lib', '$wl', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
    done
  done
done
