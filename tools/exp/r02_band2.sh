#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speckle or kitti or randomised" 2>&1 | tail -3
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
TAG="kitti" run
TAG="ref640" run --workload ref640
TAG="uhd" run --workload uhd --steps 30
TAG="fhd" run --workload fhd --steps 30
TAG="kitti1" run --pairs 1
