#!/usr/bin/env python3
"""round 4 scratch: where a border wavefront's row goes -- shader-clock cycles per phase of its row loop (development build
lib/libsbm_hip_dev.so, tools/exp/r04_devlib.sh), alone on the chip (SBM_DEV_BORDER_ONLY=1) and under the strips.
usage: SBM_LIB_AB=libsbm_hip_dev.so [SBM_DEV_BORDER_ONLY=1] python tools/exp/r04_bwprof.py [workload ...]"""
import ctypes
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import _pkg  # noqa: E402

pkg = _pkg.load()
from u96_slam_amd import synth  # noqa: E402
import torch  # noqa: E402

WL = {"kitti": (1242, 375, 128, 15, 64), "ref640": (640, 480, 64, 21, 64), "fhd": (1920, 1080, 256, 21, 16), "uhd": (3840, 2160, 256, 21, 4)}
L = pkg.load_library()
L.sbm_dev_bw_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
for wl in (sys.argv[1:] or ["kitti", "ref640"]):
    W, H, nd, w, B = WL[wl]
    Lh, Rh = synth.make_batch(0, min(B, 8), W, H, nd)
    Lh = np.concatenate([Lh] * (B // len(Lh)))[:B]; Rh = np.concatenate([Rh] * (B // len(Rh)))[:B]
    dL, dR = torch.from_numpy(Lh).cuda(), torch.from_numpy(Rh).cuda()
    bm = pkg.StereoBM.create(nd, w, device=0)
    bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32)
    for _ in range(3):
        bm.compute_device(dL, dR)
    buf = (ctypes.c_ulonglong * 8)()
    L.sbm_dev_bw_prof(buf)
    n = 10
    for _ in range(n):
        bm.compute_device(dL, dR)
    L.sbm_dev_bw_prof(buf)
    rows = max(1, buf[7])
    names = ["wait rows", "phases", "flush+stage", "winner", "finish"]
    per = [buf[i] / rows for i in range(5)]
    print(wl, "wavefront-rows", rows // n, "cycles/row:", ", ".join(f"{a} {b:.0f}" for a, b in zip(names, per)), f"total {sum(per):.0f}", flush=True)
