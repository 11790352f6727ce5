#!/bin/bash
# round 4: build lib/libsbm_hip_dev.so = the current tree's sbm_sad_fast.hip with -DSBM_DEV -DSBM_DEV_FEW (windows 15 and 21 only,
# ~30 s) linked with the other objects of the last `make`; selected at run time with SBM_LIB_AB=libsbm_hip_dev.so
set -e
cd "$(dirname "$0")/../.."
C=u96-slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBM_DEV -DSBM_DEV_FEW $EXTRA -c $C/sbm_sad_fast.hip -o /tmp/sad_fast_dev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o u96-slam_amd/lib/${OUT:-libsbm_hip_dev.so} $C/sbm_api.o $C/sbm_prefilter.o $C/sbm_sad_generic.o /tmp/sad_fast_dev.o \
  $C/sbm_sad_fast_pp.o $C/sbm_lrcheck.o $C/sbm_speckle.o $C/sbm_consume.o $C/sbm_rectify.o $C/sbm_fpga.o $C/sbm_gftt.o
