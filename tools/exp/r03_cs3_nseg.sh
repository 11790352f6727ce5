#!/bin/bash
# row-segment count of the interior kernel (SBM_FAST_NSEG) with and without column-stride-3 strips (SBM_FAST_CS3)
for w in ${WORKLOADS:-kitti ref640 fhd uhd}; do
for cs in 0 1; do
for ns in ${NSEGS:-3 4 5 6 7 8 9 10 12}; do
 SBM_FAST_CS3=$cs SBM_FAST_NSEG=$ns python bench.py --workload $w --steps 30 --warmup 8 --cpu-sample 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); s=j['roofline']['stage_ms']; print('$w cs3=$cs nseg=$ns', j['ms_per_step'], j.get('ms_per_step_median'), 'sad', s['sad'], 'border', s['border'], 'sum', round(s['sad']+s['border'],4))"
done; done; done
