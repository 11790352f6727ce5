#!/bin/bash
# round 5 experiment: windows 29 / 31 inside the interior kernel (1-column sums, 29 / 31 terms; lib/libsbm_hip_w31.so from
# tools/exp/r05_devlib.sh with FEWSET=-DSBM_DEV_FEW31) against the sliding-sum kernel (product library): parity on small
# shapes, then bench shapes, every line checked against the oracle
mkdir -p gpurun_out
SBM_LIB_AB=libsbm_hip_w31.so python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import _pkg, sbm_oracle as oracle
pkg = _pkg.load()
from test_gpu_parity import assert_stages_equal, run_engine
from u96_slam_amd import synth
bad = 0
for i, (w, h, n, nd, wsz, uniq, lr) in enumerate([(400, 100, 2, 64, 29, 10, 1), (420, 110, 1, 128, 31, 15, 1), (360, 90, 3, 32, 29, 0, -1), (700, 80, 1, 256, 31, 10, 1),
                                                   (500, 96, 2, 96, 31, 10, 1), (640, 120, 1, 192, 29, 10, 1), (900, 90, 1, 400, 29, 10, 1), (380, 100, 9, 48, 31, 5, 0),
                                                   (640, 480, 1, 64, 31, 10, 1), (1242, 375, 2, 128, 29, 10, 1)]):
    kw = dict(num_disparities=nd, block_size=wsz, prefilter_cap=31, texture_threshold=10, uniqueness_ratio=uniq, speckle_window_size=20,
              speckle_range=8, disp12_max_diff=lr)
    L, R = synth.make_batch(90 + i, n, w, h, min(nd, w // 3))
    try:
        eng, ref = run_engine(pkg, oracle, kw, L, R)
        assert_stages_equal(eng, ref, kw)
        assert np.array_equal(eng["disp"], ref["disp"])
        bm = pkg.StereoBM.create(nd, wsz); bm.compute(L[:1], R[:1])
        print("ok", (w, h, n, nd, wsz), bm.last_kernel())
    except AssertionError as e:
        bad += 1; print("MISMATCH", (w, h, n, nd, wsz), str(e)[:200])
print("mismatches", bad)
PY
for x in "--workload kitti --block 29" "--workload kitti --block 31" "--workload ref640 --block 31" "--workload fhd --block 29" "--workload kitti --block 27"; do
  for lib in libsbm_hip.so libsbm_hip_w31.so; do
    [ "$x" = "--workload kitti --block 27" ] && [ $lib = libsbm_hip_w31.so ] && continue
    SBM_LIB_AB=$lib python3 bench.py --check --cpu-sample 1 --steps 10 --warmup 2 $x 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=r['stage_ms']
print('$lib', '$x', 'ms/step', j['ms_per_step'], 'sad', s['sad'], 'check', j['cpu_baseline'].get('bit_exact_vs_gpu'), r.get('kernel'))"
  done
done
