#!/bin/bash
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
TAG="kitti b1" run --pairs 1 --steps 300
TAG="ref640 b1" run --workload ref640 --pairs 1 --steps 300
TAG="ref640 b1 nopost" run --workload ref640 --pairs 1 --steps 300 --no-postfilter
TAG="kitti b4" run --pairs 4 --steps 200
python3 tools/bench_host.py 2>/dev/null | tail -3
