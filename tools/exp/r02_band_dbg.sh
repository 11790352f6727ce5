#!/bin/bash
SBM_SPECKLE_BAND=0 timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kitti_shape_batch" 2>&1 | tail -15
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speckle_band" 2>&1 | tail -30
