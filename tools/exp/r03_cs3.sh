#!/bin/bash
# A/B of the column-stride-3 strips of the interior kernel: SBM_FAST_CS3 = 0 | 1 on the bench workloads (one process each)
for w in ${WORKLOADS:-kitti ref640 fhd uhd}; do
for cs in 0 1; do
 SBM_FAST_CS3=$cs python bench.py --workload $w --steps 40 --warmup 10 --cpu-sample 2 --check 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('$w cs3=$cs', j['ms_per_step'], j.get('ms_per_step_median'), j['roofline']['kernel'], j['roofline']['stage_ms'], 'check', j.get('check'))"
done; done
