#!/bin/bash
# round 5: full launches -- forced segment counts (SBM_FAST_NSEG, development build) around the automatic choice (0)
for spec in "kitti 64" "ref640 64" "fhd 16" "uhd 4" "fhd 64" "uhd 32"; do set -- $spec; wl=$1; np=$2
  for ns in 0 6 8 10 12 14 16 20; do
    SBM_FAST_NSEG=$ns SBM_LIB_AB=libsbm_hip_dev.so python3 bench.py --workload $wl --pairs $np --no-cpu-baseline --steps 30 --prewarm-s 0.2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=$np nseg=$ns', d['ms_per_step'], d['roofline']['stage_ms']['sad'])"
  done
done
