#!/bin/bash
# speckle band walk: parity of every variant, then stage times per band height (0 = separate runs + merge kernels)
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speckle or kitti or randomised" 2>&1 | tail -5
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for b in 0 2 4 8; do export SBM_SPECKLE_BAND=$b
TAG="band$b kitti" run
TAG="band$b ref640" run --workload ref640
TAG="band$b uhd" run --workload uhd --steps 30
TAG="band$b kitti1" run --pairs 1
done
