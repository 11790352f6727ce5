#!/bin/bash
# late re-check of two knobs on the final round-2 kernels: prefilter row tile, speckle band height
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());s=j['roofline']['stage_ms'];print('$TAG', j['ms_per_step'],s['prefilter'],s['speckle'])"; }
for r in 4 8; do export SBM_PF_ROWS=$r; TAG="pfrows$r" run; done; unset SBM_PF_ROWS
for b in 2 4; do export SBM_SPECKLE_BAND=$b; TAG="band$b kitti" run; TAG="band$b ref640" run --workload ref640; TAG="band$b fhd" run --workload fhd --steps 30; done
