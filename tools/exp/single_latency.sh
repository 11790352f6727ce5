#!/bin/bash
for wl in ref640 kitti fhd; do
  python3 bench.py --workload $wl --pairs 1 --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl n=1', d['ms_per_step'], d['roofline']['stage_ms'])"
done
