#!/bin/bash
# round 6: LDS-local band walk -- correctness (tests + repeated maps), per-kernel trace, stage times
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "speckle or reference_pair or kitti_shape" 2>&1 | tail -4 | tee $O/local_tests.txt
for b in 2 4; do SBM_SPECKLE_BAND=$b timeout 600 python3 tools/exp/r06_spk_reps.py 5 2>&1 | tail -16; done | tee $O/local_reps.txt
bash tools/exp/r06_trace.sh "loc1 libsbm_hip.so kitti 64" "loc2 libsbm_hip.so ref640 64" "loc3 libsbm_hip.so kitti 1" 2>&1 | grep -i "speckle\|lrcheck\|==" | tee $O/local_trace.txt
TAG=loc LIB=libsbm_hip.so STEPS=40 bash tools/exp/r06_base.sh
