#!/bin/bash
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for pl in 1 0; do export SBM_FAST_UNIQ_PLAIN=$pl
TAG="plain$pl kitti" run
TAG="plain$pl kitti w21" run --block 21
TAG="plain$pl ref640" run --workload ref640
TAG="plain$pl fhd" run --workload fhd --steps 30
done
unset SBM_FAST_UNIQ_PLAIN
SBM_FAST_NSTRIP=2 TAG="nstrip2 kitti" run
SBM_FAST_NSTRIP=2 TAG="nstrip2 kitti w21" run --block 21
