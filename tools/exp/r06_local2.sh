#!/bin/bash
# round 6: band walk with column segments -- parity, then automatic choice against forced S = 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "speckle or reference_pair or kitti_shape" 2>&1 | tail -2 | tee $O/local_tests.txt
for sg in 0 1 4; do SBM_SPECKLE_SEG=$sg timeout 600 python3 tools/exp/r06_spk_reps.py 3 2>&1 | grep -v "\[0, 0, 0\]\|amdgpu.ids" | sed "s/^/seg=$sg /"; done | tee $O/local_reps.txt
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$1', 'ms/step', j['ms_per_step'], 'speckle', round(s['speckle'],4), 'lr', round(s['lrcheck'],4))"; }
for spec in "kitti 64" "kitti 32" "kitti 16" "kitti 8" "kitti 4" "kitti 2" "kitti 1" "ref640 64" "ref640 16" "ref640 4" "ref640 1" "fhd 16" "fhd 4" "fhd 1" "uhd 4" "uhd 1"; do
  set -- $spec
  for sg in 0 1 2 4; do
    SBM_SPECKLE_SEG=$sg python3 bench.py --no-cpu-baseline --workload $1 --pairs $2 --steps 40 --warmup 5 2>/dev/null | line "$1x$2 seg=$sg"
  done
done | tee $O/segsweep2.txt
