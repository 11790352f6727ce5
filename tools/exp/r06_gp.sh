#!/bin/bash
# round 6: streamlined general path of the band walk against the build before (libsbm_hip_old.so), one box, alternating
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "speckle or reference_pair or kitti_shape" 2>&1 | tail -2
for sg in 0 1 4; do SBM_SPECKLE_SEG=$sg timeout 600 python3 tools/exp/r06_spk_reps.py 3 2>&1 | grep -v "\[0, 0, 0\]\|amdgpu.ids" | sed "s/^/seg=$sg /"; done
for rep in 1 2; do for lib in libsbm_hip_old.so libsbm_hip.so; do LIB=$lib bash tools/exp/r06_q.sh 2>&1 | grep -v "^==\|speckle_seam\|speckle_count\|speckle_apply" | sed "s/^/$lib /"; done; done | tee $O/gp.txt
