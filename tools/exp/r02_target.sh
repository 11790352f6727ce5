#!/bin/bash
# workgroup-count target of the interior kernel's row segmentation (SBM_FAST_TARGET) across the workloads; objective = ms per step
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['sad'], j['roofline']['stage_ms']['border'])"; }
for t in 9000 7000 5600 4500; do export SBM_FAST_TARGET=$t
TAG="t$t kitti" run
TAG="t$t kitti w9" run --block 9
TAG="t$t kitti w21" run --block 21
TAG="t$t ref640" run --workload ref640
TAG="t$t fhd" run --workload fhd --steps 30
TAG="t$t uhd" run --workload uhd --steps 30
done
