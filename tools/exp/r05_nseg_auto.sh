#!/bin/bash
# round 5: the automatic segment rule over batch sizes (development build = the tree's kernel), each checked against the oracle
for wl in kitti ref640 fhd uhd; do for np in 1 2 4 8 16 32 64; do
  [ $wl = uhd ] && [ $np -gt 32 ] && continue
  SBM_LIB_AB=${LIB:-libsbm_hip_dev.so} python3 bench.py --workload $wl --pairs $np --check --cpu-sample 1 --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=$np', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'), d['roofline'].get('kernel'))"
done; done
