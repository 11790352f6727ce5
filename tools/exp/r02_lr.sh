#!/bin/bash
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['lrcheck'])"; }
for b in 256 128 64; do export SBM_LR_BS=$b
TAG="lrbs$b kitti" run
TAG="lrbs$b ref640" run --workload ref640
TAG="lrbs$b fhd" run --workload fhd --steps 30
TAG="lrbs$b uhd" run --workload uhd --steps 20
done
SBM_LR_BS=64 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
SBM_LR_BS=128 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
