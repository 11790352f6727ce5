#!/bin/bash
# round 4: knobs of the LDS-direct interior kernel -- exchange chunk (build variants lib/libsbm_hip_XCH*.so from
# EXTRA=-DSBM_FAST_XCH128=4 OUT=... tools/exp/r04_devlib.sh), segmentation target, border chain length (development library)
one() {  # label workload lib
  SBM_LIB_AB=$3 python3 bench.py --check --cpu-sample 8 --workload $2 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$1', '$2', 'ms/step', j['ms_per_step'], 'median', j.get('ms_per_step_median'), 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
}
for rep in 1 2; do
  for wl in kitti fhd; do one xch128=2 $wl libsbm_hip_dev.so; one xch128=4 $wl libsbm_hip_XCH128_4.so; one xch128=8 $wl libsbm_hip_XCH128_8.so; done
  one xch64=4 ref640 libsbm_hip_dev.so; one xch64=2 ref640 libsbm_hip_XCH64_2.so; one xch64=8 ref640 libsbm_hip_XCH64_8.so
done
for t in 8000 16000 24000 40000; do SBM_FAST_TARGET=$t one target=$t kitti libsbm_hip_dev.so; SBM_FAST_TARGET=$t one target=$t ref640 libsbm_hip_dev.so; done
for b in 10 13 16 20 26; do SBM_DEV_BSEG=$b one bseg=$b ref640 libsbm_hip_dev.so; done
for b in 32 48 64 96; do SBM_DEV_BSEG=$b one bseg=$b kitti libsbm_hip_dev.so; done
