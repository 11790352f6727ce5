#!/bin/bash
# halo sharing (SBM_FAST_NSTRIP=2, default) against the one-strip kernel: parity first, then speed
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
for ns in 1 2; do export SBM_FAST_NSTRIP=$ns
TAG="nstrip$ns kitti" run
TAG="nstrip$ns kitti w21" run --block 21
TAG="nstrip$ns kitti w9" run --block 9
TAG="nstrip$ns ref640" run --workload ref640
TAG="nstrip$ns fhd" run --workload fhd --steps 30
done
