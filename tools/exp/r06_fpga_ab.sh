#!/bin/bash
# round 6: FPGA-flavour matcher builds against each other (tools/bench_frontend.py, vga line only), parity first
timeout 600 python -m pytest tests/test_gpu_fpga.py -m gpu -q -x 2>&1 | tail -2
for r in 1 2; do
for lib in ${LIBS:-libsbm_hip_fold.so libsbm_hip_fw2.so libsbm_hip.so}; do
  SBM_LIB_AB=$lib python3 tools/bench_frontend.py --pairs 64 --reps 20 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=j[0]['vga']['fpga_bm_w21_nd64']; print('$lib', v)"
done; done
