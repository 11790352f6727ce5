#!/bin/bash
# round 6: the refactored interior kernel (and the compressed-fatbin build) against the build before the split
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_fallback.py -m gpu -q -x 2>&1 | tail -3
for lib in libsbm_hip_z.so; do SBM_LIB_AB=$lib python3 -c "
import time,sys
sys.path.insert(0,'.'); import _pkg; t0=time.perf_counter(); pkg=_pkg.load(); L=pkg.load_library(); t1=time.perf_counter()
import numpy as np
from u96_slam_amd import synth
l,r=synth.make_batch(0,1,640,480,64); bm=pkg.StereoBM.create(64,21); t2=time.perf_counter(); d=bm.compute(l,r); t3=time.perf_counter()
print('$lib load %.1f ms, create %.1f ms, first compute %.1f ms' % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))"; done
SBM_LIB_AB=libsbm_hip.so python3 -c "
import time,sys
sys.path.insert(0,'.'); import _pkg; t0=time.perf_counter(); pkg=_pkg.load(); L=pkg.load_library(); t1=time.perf_counter()
import numpy as np
from u96_slam_amd import synth
l,r=synth.make_batch(0,1,640,480,64); bm=pkg.StereoBM.create(64,21); t2=time.perf_counter(); d=bm.compute(l,r); t3=time.perf_counter()
print('plain load %.1f ms, create %.1f ms, first compute %.1f ms' % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))"
LIBS="libsbm_hip_prev.so libsbm_hip.so libsbm_hip_z.so" WLS="kitti ref640 fhd uhd" ROUNDS=2 bash tools/exp/r05_ab.sh
