#!/bin/bash
# round 5: one pair per call -- row segments of the interior kernel (SBM_FAST_NSEG / SBM_DEV_SMALL_ROWS, development build) against the SAD stage time
for wl in ref640 kitti fhd; do for ns in 24 32 40 48 56 64; do
  SBM_FAST_NSEG=$ns SBM_LIB_AB=libsbm_hip_dev.so python3 bench.py --workload $wl --pairs 1 --check --cpu-sample 1 --steps 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=1 nseg=$ns', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'), d['roofline'].get('kernel'))"
done; done
for wl in ref640 kitti fhd; do for sr in 4 6 8 12; do for np in 1 2 4; do
  SBM_DEV_SMALL_ROWS=$sr SBM_LIB_AB=libsbm_hip_dev.so python3 bench.py --workload $wl --pairs $np --check --cpu-sample 1 --steps 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=$np small_rows=$sr', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'))"
done; done; done
