#!/bin/bash
# round 6: persistent prefilter (workgroups loop over tiles, next tile's loads in flight) -- workgroups per CU; parity first
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_frontend.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
for w in 2 3 4 6 8; do
  SBM_PF_WGS=$w python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs=$w kitti', j['ms_per_step'], j['roofline']['stage_ms']['prefilter'], j['roofline_prefilter']['frac'])"
  SBM_PF_WGS=$w python bench.py --no-cpu-baseline --steps 60 --workload ref640 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs=$w ref640', j['ms_per_step'], j['roofline']['stage_ms']['prefilter'])"
  SBM_PF_WGS=$w python bench.py --no-cpu-baseline --steps 60 --workload kitti --pairs 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs=$w kitti1', j['ms_per_step'], j['roofline']['stage_ms']['prefilter'])"
  SBM_PF_WGS=$w python3 tools/bench_prefilter.py --cold --reps 20 | cut -c100-400
done
