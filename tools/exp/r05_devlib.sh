#!/bin/bash
# round 5: lib/$OUT (default libsbm_hip_dev.so) = the tree's sbm_sad_fast.hip with -DSBM_DEV [-DSBM_DEV_FEW unless FULL=1] $EXTRA,
# linked with the other objects of the last `make`; selected at run time with SBM_LIB_AB=<name>. Several may build in parallel.
set -e
cd "$(dirname "$0")/../.."
C=u96-slam_amd/csrc
OUT=${OUT:-libsbm_hip_dev.so}
O=/tmp/sad_fast_${OUT%.so}.o
FEW=-DSBM_DEV_FEW; [ "${FULL:-0}" = 1 ] && FEW=; [ -n "$FEWSET" ] && FEW=$FEWSET
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSBM_DEV $FEW $EXTRA -c $C/sbm_sad_fast.hip -o $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o u96-slam_amd/lib/$OUT $C/sbm_api.o $C/sbm_prefilter.o $C/sbm_sad_generic.o $C/sbm_sad_wide.o $O \
  $C/sbm_sad_fast_pw1.o $C/sbm_sad_fast_pw2.o $C/sbm_sad_fast_pw3.o $C/sbm_sad_fast_pp.o $C/sbm_lrcheck.o $C/sbm_speckle.o $C/sbm_consume.o $C/sbm_rectify.o $C/sbm_fpga.o $C/sbm_gftt.o
