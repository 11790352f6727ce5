#!/bin/bash
# border segment length at the reference's call-site configuration (640x480, nd 64, w 21, 64 pairs per step)
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for sg in 0 8 12 16 24 32 48; do export SBM_BORDER_SEG=$sg
TAG="bseg$sg ref640" run --workload ref640
done
for sg in 0 8 16 24; do export SBM_BORDER_SEG=$sg
TAG="bseg$sg kitti" run
done
