#!/bin/bash
# round 5: lib/libsbm_hip_devapi.so = sbm_api.hip with -DSBM_DEV (host-phase stamps) linked with the other objects of the last `make`
set -e
cd "$(dirname "$0")/../.."
C=u96-slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSBM_DEV $EXTRA -c $C/sbm_api.hip -o /tmp/sbm_api_dev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o u96-slam_amd/lib/${OUT:-libsbm_hip_devapi.so} /tmp/sbm_api_dev.o $C/sbm_prefilter.o $C/sbm_sad_generic.o $C/sbm_sad_wide.o $C/sbm_sad_fast.o \
  $C/sbm_sad_fast_pw1.o $C/sbm_sad_fast_pw2.o $C/sbm_sad_fast_pw3.o $C/sbm_sad_fast_pp.o $C/sbm_lrcheck.o $C/sbm_speckle.o $C/sbm_consume.o $C/sbm_rectify.o $C/sbm_fpga.o $C/sbm_gftt.o
