#!/bin/bash
for w in 5 7 9 11 13 15 17 19 21 23 25 27; do
  python3 bench.py --block $w --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kitti w=$w', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline']['stage_ms']['border'], d['roofline']['kernel'])"
done
