#!/bin/bash
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for bf in 1 0; do export SBM_BORDER_FIRST=$bf
TAG="bfirst$bf kitti" run
TAG="bfirst$bf kitti w21" run --block 21
TAG="bfirst$bf ref640" run --workload ref640
TAG="bfirst$bf fhd" run --workload fhd --steps 30
TAG="bfirst$bf uhd" run --workload uhd --steps 20
TAG="bfirst$bf kitti b8" run --pairs 8
TAG="bfirst$bf kitti b1" run --pairs 1
done
