#!/bin/bash
# round 5: software-pipelined exchange (SBM_FAST_XPIPE: bit 0 two-level sums, bit 1 direct sums) against the chunk-by-chunk one;
# dev builds lib/libsbm_hip_xp{0,1,3}.so (tools/exp/r05_devlib.sh, windows 15 / 21). Every line is checked against the oracle.
LIBS="libsbm_hip_xp0.so libsbm_hip_xp1.so libsbm_hip_xp3.so" WLS="kitti ref640 fhd uhd" ROUNDS=2 bash tools/exp/r05_ab.sh
