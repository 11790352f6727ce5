#!/bin/bash
# one workgroup barrier per row in the SAD kernel (second exchange read one row late): parity, then stage times
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
TAG="kitti" run
TAG="kitti w21" run --block 21
TAG="kitti w9" run --block 9
TAG="ref640" run --workload ref640
TAG="fhd" run --workload fhd --steps 30
TAG="uhd" run --workload uhd --steps 30
TAG="kitti1" run --pairs 1
