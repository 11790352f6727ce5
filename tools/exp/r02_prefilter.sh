#!/bin/bash
for r in 4 2 8; do echo "PF_ROWS=$r"; SBM_PF_ROWS=$r python3 tools/bench_prefilter.py --reps 20; 
SBM_PF_ROWS=$r python3 bench.py --no-cpu-baseline | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('engine', j['ms_per_step'],j['roofline_prefilter'])"
done
