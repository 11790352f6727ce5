#!/bin/bash
# overlap mode (prefilter of call k+1 under the post-filters of call k): parity, then the bench with and without it
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for ov in 1 0; do export SBM_BENCH_OVERLAP=$ov
TAG="overlap$ov kitti" run
TAG="overlap$ov kitti" run
TAG="overlap$ov ref640" run --workload ref640
TAG="overlap$ov fhd" run --workload fhd --steps 30
TAG="overlap$ov uhd" run --workload uhd --steps 30
done
