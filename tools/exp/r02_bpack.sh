#!/bin/bash
# border kernel: 2 / 4 row segments per wavefront when few disparities leave lanes idle -- parity, then stage times
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for ns in 1 0; do export SBM_BORDER_NSUB=$ns
TAG="nsub$ns ref640" run --workload ref640
TAG="nsub$ns ref640 w9" run --workload ref640 --block 9
TAG="nsub$ns kitti" run
done
