#!/bin/bash
# round 4: build lib/libsbm_hip_devpost.so = the current tree's sbm_lrcheck.hip with -DSBM_DEV (the LR kernel's block / pixels-per-
# thread knobs SBM_DEV_LR_BS, SBM_DEV_LR_PX) linked with the other objects of the last `make`; SBM_LIB_AB=libsbm_hip_devpost.so
set -e
cd "$(dirname "$0")/../.."
C=u96-slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBM_DEV $EXTRA -Iinclude -c $C/sbm_lrcheck.hip -o /tmp/lrcheck_dev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o u96-slam_amd/lib/${OUT:-libsbm_hip_devpost.so} $C/sbm_api.o $C/sbm_prefilter.o $C/sbm_sad_generic.o $C/sbm_sad_fast.o \
  $C/sbm_sad_fast_pp.o /tmp/lrcheck_dev.o $C/sbm_consume.o $C/sbm_rectify.o $C/sbm_fpga.o $C/sbm_gftt.o
