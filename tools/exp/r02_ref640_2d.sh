#!/bin/bash
# 640x480 nd64 w21 x64: interior segment count x border segment length (the border kernel sets the length of the SAD stage here)
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['sad'], j['roofline']['stage_ms']['border'])"; }
for ns in 0 7 9; do for sg in 8 12 16 24 32; do
export SBM_FAST_NSEG=$ns SBM_BORDER_SEG=$sg
TAG="nseg$ns bseg$sg" run --workload ref640
done; done
