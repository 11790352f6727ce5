#!/bin/bash
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms']['sad'],j['roofline']['stage_ms']['border'])"; }
for t in 5000 7000 9000 11000 14000 18000; do SBM_FAST_TARGET=$t TAG="target$t" run; done
for s in 6 8 16 24; do SBM_BORDER_SEG=$s TAG="bseg$s" run; done
for t in 0 1; do SBM_FAST_TAPER=$t TAG="taper$t" run; done
