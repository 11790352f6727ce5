#!/bin/bash
for cfg in "1 7" "1 9" "0 9" "1 5"; do set -- $cfg
  for rep in 1 2; do
  SBM_FAST_TAPER=$1 SBM_FAST_NSEG=$2 python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('taper=$1 nseg=$2', d['ms_per_step'], d['roofline']['stage_ms'])"
  done
done
