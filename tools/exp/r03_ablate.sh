#!/bin/bash
# Builds lib/libsbm_hip_abl{1..6}.so: the engine with ONE phase of the SAD row loop left out (SBM_ABL, see
# sbm_sad_fast.hip) -- run in the build container; then `tools/exp/r03_ablate.sh run` on the GPU box prints the SAD stage
# time of each (marginal cost of the phase = full - ablated). Results of ablated builds are wrong by construction.
set -e
cd "$(dirname "$0")/../.."
if [ "${1:-build}" = build ]; then
  for n in 1 2 3 4 5 6; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSBM_ABL=$n -c u96-slam_amd/csrc/sbm_sad_fast.hip -o /tmp/sad_fast_abl$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o u96-slam_amd/lib/libsbm_hip_abl$n.so /tmp/sad_fast_abl$n.o \
        $(ls u96-slam_amd/csrc/*.o | grep -v sbm_sad_fast.o) ) &
  done
  wait
  ls -la u96-slam_amd/lib/
else
  for wl in ${WLS:-kitti ref640}; do
    for n in 0 1 2 3 4 5 6 0; do
      lib=libsbm_hip_abl$n.so; [ $n = 0 ] && lib=libsbm_hip.so
      SBM_LIB_AB=$lib python3 bench.py --no-cpu-baseline --workload $wl --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms']
print('abl$n', '$wl', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'border', s['border'])"
    done
  done
fi
