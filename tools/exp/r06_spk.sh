#!/bin/bash
# round 6: speckle rework -- parity of the variants, then stage times against the round-5 build and across segment counts
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "speckle or lr_check or border_columns" 2>&1 | tail -5 | tee $O/spk_tests.txt
for seg in 1 2 3 5 8; do
  SBM_SPECKLE_SEG=$seg timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "speckle" 2>&1 | tail -2 | sed "s/^/seg=$seg /" | tee -a $O/spk_tests.txt
done
TAG=r05 LIB=libsbm_hip_r05.so STEPS=40 bash tools/exp/r06_base.sh
TAG=new LIB=libsbm_hip.so STEPS=40 bash tools/exp/r06_base.sh
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=r['stage_ms']
print('$1', 'ms/step', j['ms_per_step'], 'speckle', round(s['speckle'],4), 'lr', round(s['lrcheck'],4))"; }
for spec in "kitti 64" "ref640 64" "kitti 1" "ref640 1" "uhd 4"; do
  set -- $spec
  for seg in 1 2 3 4 6 8; do
    for band in 2 4; do
      SBM_SPECKLE_SEG=$seg SBM_SPECKLE_BAND=$band python3 bench.py --no-cpu-baseline --workload $1 --pairs $2 --steps 40 --warmup 5 2>/dev/null | line "$1x$2 seg=$seg band=$band"
    done
  done
done | tee $O/spk_sweep.txt
