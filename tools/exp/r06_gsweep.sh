#!/bin/bash
# round 6: band height of the speckle band walk across launch sizes (stage times from bench.py)
O=gpurun_out/r06; mkdir -p $O
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$1', 'ms/step', j['ms_per_step'], 'speckle', round(s['speckle'],4), 'lr', round(s['lrcheck'],4))"; }
for spec in "kitti 64" "kitti 32" "kitti 16" "kitti 8" "kitti 1" "ref640 64" "ref640 16" "ref640 1" "fhd 16" "fhd 1" "uhd 4"; do
  set -- $spec
  for band in 2 4; do
    SBM_SPECKLE_BAND=$band python3 bench.py --no-cpu-baseline --workload $1 --pairs $2 --steps 40 --warmup 5 2>/dev/null | line "$1x$2 band=$band"
  done
done | tee $O/gsweep.txt
