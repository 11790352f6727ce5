#!/bin/bash
# round 5: 257 .. 512 disparities inside the interior kernel (<128,3>, <128,4>; border columns from the sliding-sum kernel):
# parity tests, then ms per step at frame sizes. usage (GPU box): bash tools/exp/r05_nd512.sh
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_wide.py -x -q -m gpu 2>&1 | tail -5
for x in "--workload kitti --ndisp 512 --pairs 16" "--workload kitti --ndisp 384 --pairs 16" "--workload fhd --ndisp 512 --pairs 16" "--workload fhd --ndisp 320 --pairs 16" "--workload uhd --ndisp 512 --pairs 4" "--workload uhd --ndisp 512 --pairs 32" "--workload fhd --ndisp 512 --block 19 --pairs 16" "--workload uhd --ndisp 512 --pairs 1"; do
    python3 bench.py --check --cpu-sample 1 --steps 10 --warmup 2 $x 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$x', 'ms/step', j['ms_per_step'], 'stages', s, 'value', j['value'], j['unit'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
done
