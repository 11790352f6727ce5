#!/bin/bash
# round 6: LDS-local band walk -- record capacity per row (LDS per wavefront against re-walks)
O=gpurun_out/r06; mkdir -p $O
for b in 2 4; do SBM_SPECKLE_BAND=$b timeout 600 python3 tools/exp/r06_spk_reps.py 3 2>&1 | tail -1; done | tee $O/local_reps.txt
specs=()
for v in "" _capr64; do
  for wl in "kitti 64" "ref640 64" "uhd 4" "kitti 1" "fhd 16"; do set -- $wl; specs+=("ab${v}_$1$2 libsbm_hip$v.so $1 $2"); done
done
bash tools/exp/r06_trace.sh "${specs[@]}" 2>&1 | grep -i "speckle_band\|==" | tee $O/local_ab.txt
