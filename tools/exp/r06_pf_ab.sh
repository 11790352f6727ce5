#!/bin/bash
# round 6: what the prefilter's load pattern costs -- ablation builds (WRONG results by construction: the loads are moved to
# 16-byte aligned addresses / the extra dword load is dropped), in-engine stage time and the cold engine-sized launch
for lib in libsbm_hip.so libsbm_hip_pfab1.so libsbm_hip_pfab2.so libsbm_hip_pfab3.so; do
  SBM_LIB_AB=$lib python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib kitti64 prefilter stage', j['roofline']['stage_ms']['prefilter'])"
  SBM_LIB_AB=$lib python3 tools/bench_prefilter.py --cold --reps 20 | cut -c100-420
done
