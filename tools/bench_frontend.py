#!/usr/bin/env python3
"""Times the front-end kernels (rectification resampler, stand-alone prefilter) and the map consumers on cuda:0 and
prints achieved algorithmic GB/s against the 8 TB/s HBM peak. Wall clock around R back-to-back launches on the engine's
stream (each call is one kernel; launches are asynchronous, one sync at the end).

usage: python tools/bench_frontend.py [--pairs 64] [--reps 50]"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import numpy as np
    import torch
    import _pkg

    pkg = _pkg.load()
    from test_frontend import CAM_L, scaled_cam

    out = []
    for name, W, H in (("vga", 640, 480), ("kitti", 1242, 375), ("fhd", 1920, 1080)):
        n = 2 * args.pairs if W < 1900 else max(2, args.pairs // 2)
        bm = pkg.StereoBM.create(64, 21)
        L = bm._L
        rng = np.random.default_rng(1)
        src = torch.from_numpy(rng.integers(0, 256, (n, H, W), dtype=np.uint8)).cuda()
        dst = torch.empty_like(src)
        cam = pkg.make_rect_cam(**(CAM_L if name == "vga" else scaled_cam(CAM_L, W, H)))
        rmap = bm.rect_map(cam, W, H)
        disp = torch.from_numpy(rng.integers(-16, 2000, (n, H, W)).astype(np.int16)).cuda()
        xyz = torch.empty((n, H // 4, W // 4, 3), dtype=torch.float32, device="cuda")
        dec = torch.empty((n, H // 4, W // 4), dtype=torch.int16, device="cuda")
        model = pkg.StereoModel()
        model.fx_l = model.fy_l = model.fx_r = model.fy_r = 718.856
        model.cx_l = model.cx_r = 607.19; model.cy_l = 185.2; model.Tx_r = -386.1
        npx = n * W * H

        def timed(fn, bytes_):
            for _ in range(3):
                fn()
            bm.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            bm.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            return {"ms": round(ms, 4), "algorithmic_GBps": round(bytes_ / ms / 1e6, 1), "frac_of_8TBps": round(bytes_ / ms / 1e6 / 8000, 4)}

        r = {"shape": f"{n}x{H}x{W}"}
        r["rect_remap"] = timed(lambda: L.sbm_rect_remap_device(bm._h, n, src.data_ptr(), rmap.data_ptr(), W, H, dst.data_ptr(), 0), 2 * npx)
        r["rect_map"] = timed(lambda: L.sbm_rect_map_device(bm._h, ctypes.byref(cam), W, H, rmap.data_ptr(), 0), 4 * W * H)
        r["prefilter_cv"] = timed(lambda: L.sbm_prefilter_device(bm._h, n, src.data_ptr(), W, H, 0, 31, dst.data_ptr(), 0), 2 * npx)
        r["prefilter_rtl"] = timed(lambda: L.sbm_prefilter_device(bm._h, n, src.data_ptr(), W, H, 1, 31, dst.data_ptr(), 0), 2 * npx)
        r["decimate4"] = timed(lambda: L.sbm_decimate_device(bm._h, n, disp.data_ptr(), W, H, 4, dec.data_ptr(), 0), npx // 16 * 4)
        r["reproject_dec4"] = timed(lambda: L.sbm_reproject_device(bm._h, n, dec.data_ptr(), W // 4, H // 4, 4, ctypes.byref(model), 0, xyz.data_ptr(), 0), npx // 16 * 14)
        # the reference's own PL blocks: GFTT map (1 B in + 2 B out per pixel) and the FPGA-flavour
        # matcher at the firmware's configuration (window 21, 64 disparities) -- VALU-bound like the cv flavour
        if W <= 1023 and H <= 511:
            eig = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
            mx = torch.empty((n,), dtype=torch.int32, device="cuda")
            r["gftt_eig"] = timed(lambda: L.sbm_gftt_eig_device(bm._h, n, src.data_ptr(), W, H, eig.data_ptr(), mx.data_ptr(), 0), 3 * npx)
            fp = pkg.fpga_params(W, H, 21, 64)
            xs = torch.from_numpy(rng.integers(0, 64, (n, H, W), dtype=np.uint8)).cuda()
            xr = torch.roll(xs, -17, dims=2).contiguous()
            t = timed(lambda: L.sbm_fpga_bm_device(bm._h, n // 2, xs.data_ptr(), xr.data_ptr(), ctypes.byref(fp), disp.data_ptr(), 0), 4 * (n // 2) * W * H)
            t["Mpix_disparities_per_s"] = round((n // 2) * W * H * 64 / t["ms"] / 1e3, 1)
            r["fpga_bm_w21_nd64"] = t
        out.append({name: r})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
