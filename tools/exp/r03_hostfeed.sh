#!/bin/bash
# host feed (pinned, chunked three-stream pipeline): default chunk plan (8 | <=24 ... | 8) against uniform chunks
for c in 0 8 16 24 32 64; do
  SBM_HOST_CHUNK=$c python3 bench.py --feed host --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); h=j['host_feed']
print('chunk $c ms/step', j['ms_per_step'], 'min', j['ms_per_step_min'], 'legs', h['resident_compute_ms'], h['h2d_ms'], h['d2h_ms'], 'overlap', h['overlap_frac'])"
done
