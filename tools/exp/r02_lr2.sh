#!/bin/bash
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['lrcheck'])"; }
for b in 0 64 128; do export SBM_LR_BS=$b
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
TAG="bs$b kitti" run
TAG="bs$b fhd" run --workload fhd --steps 30
TAG="bs$b uhd" run --workload uhd --steps 30
TAG="bs$b ref640" run --workload ref640
TAG="bs$b kitti1" run --pairs 1
done
