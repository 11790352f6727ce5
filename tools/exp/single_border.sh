#!/bin/bash
for v in 1 2; do for bs in 0 4 6 8 12; do
  SBM_BORDER_V=$v SBM_BORDER_SEG=$bs python3 bench.py --workload ref640 --pairs 1 --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref640 n=1 v=$v bseg=$bs', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline']['stage_ms']['border'])"
done; done
