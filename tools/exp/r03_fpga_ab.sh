#!/bin/bash
# FPGA-flavour matcher: current build against lib/libsbm_hip_fpold.so (an earlier sbm_fpga.hip), alternating
for rep in 1 2 3; do
  for lib in libsbm_hip_fpold.so libsbm_hip.so; do
    SBM_LIB_AB=$lib python3 tools/bench_frontend.py 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib', j[0]['vga']['fpga_bm_w21_nd64']['ms'], j[0]['vga']['gftt_eig']['ms'])"
  done
done
