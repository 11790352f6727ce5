#!/bin/bash
# round 5: the fast envelope at the bench shape (KITTI 1242x375 x 64 pairs, all post-filters): ms per step over block sizes and
# disparity counts, each checked against the oracle on an 8-pair sample. usage: [SBM_LIB_AB=lib] [NDS="128"] bash tools/exp/r05_envelope.sh
for nd in ${NDS:-32 64 128 256}; do
  for w in ${WS:-5 7 9 11 13 15 17 19 21 23 25 27}; do
    python3 bench.py --check --cpu-sample 8 --workload kitti --block $w --ndisp $nd --steps 30 --warmup 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('nd', $nd, 'w', $w, 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'Tpd/s', round(j['value']/1e6,2), 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
  done
done
