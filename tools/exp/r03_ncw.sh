#!/bin/bash
# A/B of the column-cooperating interior kernel: SBM_FAST_NCW = 1 | 2 | 4 on the bench workloads (one process each)
for w in ${WORKLOADS:-kitti ref640}; do
for ncw in 1 2 4; do
 SBM_FAST_NCW=$ncw python bench.py --workload $w --steps 40 --warmup 10 --cpu-sample 2 --check 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('$w ncw=$ncw', j['ms_per_step'], j.get('ms_per_step_median'), j['roofline']['kernel'], j['roofline']['stage_ms'], 'check', j.get('check'))"
done; done
