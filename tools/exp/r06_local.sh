#!/bin/bash
# round 6: LDS-local band walk with column segments -- correctness (tests + repeated maps), per-kernel trace, stage times
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "speckle or reference_pair or kitti_shape" 2>&1 | tail -4 | tee $O/local_tests.txt
for sg in 1 2 4; do for b in 2 4; do SBM_SPECKLE_SEG=$sg SBM_SPECKLE_BAND=$b timeout 600 python3 tools/exp/r06_spk_reps.py 3 2>&1 | grep -v "\[0, 0, 0\]" | sed "s/^/seg=$sg band=$b /"; done; done | tee $O/local_reps.txt
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$1', 'ms/step', j['ms_per_step'], 'speckle', round(s['speckle'],4), 'lr', round(s['lrcheck'],4))"; }
for spec in "kitti 64" "kitti 32" "kitti 8" "kitti 1" "ref640 64" "ref640 16" "ref640 1" "fhd 16" "fhd 1" "uhd 4"; do
  set -- $spec
  for band in 2 4; do for sg in 1 2 4; do
    SBM_SPECKLE_SEG=$sg SBM_SPECKLE_BAND=$band python3 bench.py --no-cpu-baseline --workload $1 --pairs $2 --steps 40 --warmup 5 2>/dev/null | line "$1x$2 band=$band seg=$sg"
  done; done
done | tee $O/segsweep.txt
