#!/bin/bash
# round 6: wavefronts per workgroup of the band walk at one segment per band (4 bands per workgroup hold their LDS until the slowest is done)
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for lib in libsbm_hip.so libsbm_hip_nw2.so libsbm_hip_nw1.so; do LIB=$lib bash tools/exp/r06_q.sh 2>&1 | grep -v "^==\|speckle_seam\|speckle_count\|speckle_apply" | sed "s/^/$lib /"; done; done | tee $O/nw.txt
