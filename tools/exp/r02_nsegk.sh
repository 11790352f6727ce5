#!/bin/bash
# row-segment count / taper of the interior kernel at the bench default (22 strips x 64 pairs x nseg workgroups of 2 wavefronts)
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['sad'], j['roofline']['stage_ms']['border'])"; }
for rep in 1 2; do
for ns in 0 2 3 4 5; do export SBM_FAST_NSEG=$ns
TAG="nseg$ns kitti" run
done
done
export SBM_FAST_NSEG=4
TAG="nseg4 kitti w21" run --block 21
TAG="nseg4 kitti w9" run --block 9
export SBM_FAST_NSEG=0
TAG="nseg0 kitti w21" run --block 21
TAG="nseg0 kitti w9" run --block 9
