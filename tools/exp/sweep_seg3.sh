#!/bin/bash
for cfg in "1 1 7" "1 1 9" "1 0 9" "1 1 5" "0 1 7" "0 0 9"; do set -- $cfg
  for rep in 1 2; do
  SBM_SIDE_SWAP=$1 SBM_FAST_TAPER=$2 SBM_FAST_NSEG=$3 python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('swap=$1 taper=$2 nseg=$3', d['ms_per_step'], d['roofline']['stage_ms'])"
  done
done
