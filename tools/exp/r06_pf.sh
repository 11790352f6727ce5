#!/bin/bash
# round 6: in-engine prefilter, rows per thread tile (development build of sbm_prefilter.hip, SBM_PF_ROWS) + the default bench line
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$1', 'ms/step', j['ms_per_step'], 'prefilter', round(s['prefilter'],4), j.get('roofline_prefilter'))"; }
for rep in 1 2; do for rows in 2 4 8; do
  SBM_LIB_AB=libsbm_hip_pfdev.so SBM_PF_ROWS=$rows python3 bench.py --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | line "kitti64 rows=$rows"
  SBM_LIB_AB=libsbm_hip_pfdev.so SBM_PF_ROWS=$rows python3 bench.py --no-cpu-baseline --steps 60 --warmup 5 --workload ref640 2>/dev/null | line "ref640x64 rows=$rows"
done; done
python3 bench.py --check 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k:j[k] for k in ('ms_per_step','value','cpu_baseline')}, indent=0))"
