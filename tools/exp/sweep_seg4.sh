#!/bin/bash
for wl in kitti fhd ref640 uhd; do for tgt in 6000 9000 12000 16000; do
  for rep in 1 2; do
  SBM_FAST_TARGET=$tgt python3 bench.py --workload $wl --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl target=$tgt', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline']['stage_ms']['border'])"
  done
done; done
