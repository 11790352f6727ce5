#!/usr/bin/env python3
"""round 5: where the host side of sbm_compute() goes for one pair per call (the reference's call pattern, main.cpp:201-216).
Needs a -DSBM_DEV build of sbm_api.hip (lib/libsbm_hip_devapi.so, tools/exp/r05_devapi.sh); phases are wall-clock stamps inside
sbm_compute_batch: 0 = device scope + staging check, 1 = the two H2D copies (submit; pageable memory: the runtime stages them),
2 = the seven kernel launches, 3 = the D2H copy (submit), 4 = hipStreamSynchronize.  usage: SBM_LIB_AB=libsbm_hip_devapi.so python3 tools/exp/r05_host_attrib.py"""
import ctypes, pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import _pkg
pkg = _pkg.load()
from u96_slam_amd import synth
lib = pkg.load_library()
for name, W, H, nd, w in (("ref640", 640, 480, 64, 21), ("kitti", 1242, 375, 128, 15)):
    L, R = synth.make_batch(0, 1, W, H, nd)
    for pinned in (0, 1):
        l, r = np.ascontiguousarray(L[0]), np.ascontiguousarray(R[0])
        disp = np.empty((H, W), np.int16)
        if pinned:
            import torch
            tl, tr, td = torch.from_numpy(l).pin_memory(), torch.from_numpy(r).pin_memory(), torch.empty((H, W), dtype=torch.int16).pin_memory()
            l, r, disp = tl.numpy(), tr.numpy(), td.numpy()
        bm = pkg.StereoBM.create(nd, w)
        bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10)
        bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
        for _ in range(20): bm.compute(l, r, disp)
        acc = (ctypes.c_double * 8)(); calls = ctypes.c_ulonglong()
        if hasattr(lib, "sbm_dev_host_prof"): lib.sbm_dev_host_prof(acc, ctypes.byref(calls))
        reps = 500
        t0 = time.perf_counter()
        for _ in range(reps): bm.compute(l, r, disp)
        ms = (time.perf_counter() - t0) / reps * 1e3
        line = f"{name} pinned={pinned} ms/call {ms:.4f}"
        if hasattr(lib, "sbm_dev_host_prof"):
            lib.sbm_dev_host_prof(acc, ctypes.byref(calls))
            line += " us/phase " + " ".join(f"{acc[i] / max(calls.value, 1):.1f}" for i in range(5)) + f" (sum {sum(acc[i] for i in range(5)) / max(calls.value, 1):.1f})"
        print(line, flush=True)
