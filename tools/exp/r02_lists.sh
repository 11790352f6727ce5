#!/bin/bash
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
SBM_SPECKLE_LISTS=0 timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['speckle'])"; }
for v in 1 0; do export SBM_SPECKLE_LISTS=$v
TAG="lists$v kitti" run
TAG="lists$v ref640" run --workload ref640
TAG="lists$v fhd" run --workload fhd --steps 30
TAG="lists$v uhd" run --workload uhd --steps 20
done
python3 tools/soak.py --iters 1500 --seed 5
