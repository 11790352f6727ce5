#!/bin/bash
# swap the built library for each variant and bench (restores the original afterwards)
cp u96-slam_amd/lib/libsbm_hip.so /tmp/orig.so
for tag in base relaxed nopostmisched; do
  cp tools/exp/libsbm_$tag.so u96-slam_amd/lib/libsbm_hip.so
  for rep in 1 2; do python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['stage_ms']['sad'])"; done
done
cp /tmp/orig.so u96-slam_amd/lib/libsbm_hip.so
