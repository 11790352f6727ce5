#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (sbm_compute / sbm_compute_batch): what a drop-in caller that
hands over cv::Mat-style host memory sees. Not bench.py's `value` (that one starts with inputs resident in HBM)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np
    import _pkg

    pkg = _pkg.load()
    from u96_slam_amd import synth

    out = {}
    for name, W, H, nd, wsz, n in (("ref640_single", 640, 480, 64, 21, 1), ("kitti_single", 1242, 375, 128, 15, 1),
                                   ("kitti_b64", 1242, 375, 128, 15, 64), ("fhd_b16", 1920, 1080, 256, 21, 16)):
        L, R = synth.make_batch(0, min(n, 4), W, H, nd)
        L = np.ascontiguousarray(np.concatenate([L] * (n // len(L))) if n > 1 else L[0])
        R = np.ascontiguousarray(np.concatenate([R] * (n // len(R))) if n > 1 else R[0])
        bm = pkg.StereoBM.create(nd, wsz)
        bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10)
        bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
        disp = np.empty(L.shape, np.int16)
        for _ in range(3):
            bm.compute(L, R, disp)
        reps = 200 if n == 1 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            bm.compute(L, R, disp)
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name] = {"ms_per_call": round(ms, 4), "ms_per_pair": round(ms / n, 4),
                     "Mpix_disp_per_s": round(n * W * H * nd / ms / 1e3, 1),
                     "host_bytes_per_call": int(n * W * H * 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
