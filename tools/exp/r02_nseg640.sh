#!/bin/bash
# row-segment count of the interior kernel at 640x480 nd64 w21 x64 (single-wavefront workgroups: 13 strips x 64 pairs x nseg)
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['sad'], j['roofline']['stage_ms']['border'])"; }
for ns in 0 6 8 9 10 11 12 13 14 16 20; do export SBM_FAST_NSEG=$ns
TAG="nseg$ns ref640" run --workload ref640
done
unset SBM_FAST_NSEG
for t in 0 1; do export SBM_FAST_TAPER=$t
TAG="taper$t ref640" run --workload ref640
done
