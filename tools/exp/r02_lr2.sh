#!/bin/bash
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
TAG="kitti" run
TAG="fhd" run --workload fhd --steps 30
TAG="uhd" run --workload uhd --steps 30
TAG="ref640" run --workload ref640
TAG="kitti1" run --pairs 1
