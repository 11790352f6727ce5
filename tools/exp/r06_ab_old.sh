#!/bin/bash
# round 6: the build before column segments (libsbm_hip_old.so) against the current one on one box, alternating
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do for lib in libsbm_hip_old.so libsbm_hip.so; do LIB=$lib bash tools/exp/r06_q.sh 2>&1 | grep -v "^==\|speckle_seam\|speckle_count\|speckle_apply" | sed "s/^/$lib /"; done; done | tee $O/ab_old.txt
python3 tools/bench_prefilter.py --cold --reps 20 | tee $O/pf_cold.json
