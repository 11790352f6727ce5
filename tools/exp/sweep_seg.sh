#!/bin/bash
# sweep row-segment count and taper of the fast SAD kernel (KITTI b64)
for taper in 0 1; do for nseg in 5 6 7 8 9 10 12; do
  r=$(SBM_FAST_TAPER=$taper SBM_FAST_NSEG=$nseg python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['stage_ms']['sad'])")
  echo "taper=$taper nseg=$nseg step/sad ms: $r"
done; done
