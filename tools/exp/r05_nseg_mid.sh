#!/bin/bash
# round 5: batches between one pair and a full chip -- forced segment counts (SBM_FAST_NSEG) against the automatic rule (0)
for wl in kitti ref640 fhd; do for np in 2 4 8 16 32; do for ns in 0 8 12 16 24 32 48; do
  SBM_FAST_NSEG=$ns SBM_LIB_AB=libsbm_hip_dev.so python3 bench.py --workload $wl --pairs $np --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=$np nseg=$ns', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline'].get('kernel'))"
done; done; done
