#!/bin/bash
for wl in kitti ref640; do for bs in 8 12 16 24 32 48 64; do
  SBM_BORDER_SEG=$bs python3 bench.py --workload $wl --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl bseg=$bs', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline']['stage_ms']['border'])"
done; done
for wl in kitti ref640; do for bv in 1 2; do
  SBM_BORDER_V=$bv python3 bench.py --workload $wl --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl bv=$bv', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline']['stage_ms']['border'])"
done; done
