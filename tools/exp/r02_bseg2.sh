#!/bin/bash
# border segment length / interior taper with the 5600-workgroup interior target, bench default
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['sad'], j['roofline']['stage_ms']['border'])"; }
for sg in 0 6 8 10 16 24; do export SBM_BORDER_SEG=$sg
TAG="bseg$sg kitti" run
done
unset SBM_BORDER_SEG
for t in 0 1; do export SBM_FAST_TAPER=$t
TAG="taper$t kitti" run
done
