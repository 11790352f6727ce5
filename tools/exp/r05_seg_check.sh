#!/bin/bash
# round 5: the segment formula (automatic) across workloads and batch sizes, each checked; LIB selects the build
for spec in "kitti 64" "ref640 64" "fhd 16" "uhd 4" "fhd 64" "uhd 32" "kitti 32" "kitti 16" "kitti 8" "kitti 4" "kitti 2" "kitti 1" "ref640 16" "ref640 8" "ref640 1" "fhd 4" "fhd 1"; do set -- $spec; wl=$1; np=$2
  for rep in 1 2; do
    SBM_LIB_AB=${LIB:-libsbm_hip_dev.so} python3 bench.py --workload $wl --pairs $np --check --cpu-sample 1 --steps 30 --prewarm-s 0.2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=$np', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'))"
  done
done
for wl in kitti; do for w in 9 27 19; do for nd in 64 256; do
  SBM_LIB_AB=${LIB:-libsbm_hip_dev.so} python3 bench.py --workload kitti --block $w --ndisp $nd --check --cpu-sample 2 --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kitti w $w nd $nd', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'))"
done; done; done
