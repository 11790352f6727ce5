#!/bin/bash
# seam / count kernels of the speckle filter: every pair on one XCD with L2-scope atomics (default) against agent scope
# (SBM_XCD_AFFINITY=0)
for w in ${WORKLOADS:-kitti ref640 fhd uhd}; do
for f in 0 1; do
 SBM_XCD_AFFINITY=$f python bench.py --workload $w --steps 40 --warmup 10 --cpu-sample 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); s=j['roofline']['stage_ms']; print('$w affinity=$f', j['ms_per_step'], j.get('ms_per_step_median'), 'lr', s['lrcheck'], 'speckle', s['speckle'])"
done; done
