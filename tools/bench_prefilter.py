#!/usr/bin/env python3
"""Prefilter kernel against the device's own copy rate, inside and beyond the 256 MB Infinity Cache.

For each working set (images in + planes out): stand-alone prefilter kernel (sbm_prefilter_device, cv flavour), a plain
device-to-device copy of the same bytes (torch copy_ = the runtime's copy kernel) and a hand-rolled 16-byte-per-lane copy
kernel are timed over back-to-back launches; algorithmic bytes = 1 B read + 1 B written per pixel.
--cold (round 6): the size of the engine's own launch (64 KITTI pairs = 119 MB) the way the engine sees it -- sources and
destination NOT in the Infinity Cache (a 1 GB copy runs between the timed launches), every launch timed by its own pair of
events -- for the prefilter and for the runtime's copy of the same bytes.
usage: python tools/bench_prefilter.py [--reps 30] [--cold]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--cold", action="store_true")
    args = ap.parse_args()
    import torch
    import _pkg

    pkg = _pkg.load()
    bm = pkg.StereoBM.create(64, 21)
    L = bm._L
    out = []
    if args.cold:
        W, H, n = 1242, 375, 128            # 128 images = the left and right images of 64 pairs
        src = torch.randint(0, 256, (n, H, W), dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        big_a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        big_b = torch.empty_like(big_a)
        nbytes = 2.0 * n * W * H
        stream = torch.cuda.ExternalStream(int(L.sbm_stream(bm._h)))   # the engine launches on its own stream

        def cold(fn, on_engine_stream):
            ms = []
            for i in range(args.reps + 3):
                big_b.copy_(big_a)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if on_engine_stream:
                    with torch.cuda.stream(stream):
                        e0.record(); fn(); e1.record()
                    bm.synchronize()
                else:
                    e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                if i >= 3:
                    ms.append(e0.elapsed_time(e1))
            ms.sort()
            med = ms[len(ms) // 2]
            return {"ms_median": round(med, 4), "ms_min": round(ms[0], 4), "GBps": round(nbytes / med / 1e6, 1),
                    "frac_of_8TBps": round(nbytes / med / 1e6 / 8000, 4)}

        r = {"workload": "kitti x128 images (119 MB), cold: a 1 GB copy between launches, one launch per event pair"}
        r["device_copy"] = cold(lambda: dst.copy_(src), False)
        r["prefilter"] = cold(lambda: L.sbm_prefilter_device(bm._h, n, src.data_ptr(), W, H, 0, 31, dst.data_ptr(), 0), True)
        r["prefilter_vs_copy"] = round(r["prefilter"]["GBps"] / r["device_copy"]["GBps"], 3)
        print(json.dumps([r]))
        return
    for name, W, H, n in (("kitti x128 (119 MB)", 1242, 375, 128), ("kitti x512 (477 MB)", 1242, 375, 512),
                          ("kitti x2048 (1.9 GB)", 1242, 375, 2048), ("fhd x512 (2.1 GB)", 1920, 1080, 512)):
        src = torch.randint(0, 256, (n, H, W), dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        nbytes = 2.0 * n * W * H

        def timed(fn, sync):
            for _ in range(3):
                fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            sync()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            return {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4)}

        r = {"workload": name}
        r["prefilter"] = timed(lambda: L.sbm_prefilter_device(bm._h, n, src.data_ptr(), W, H, 0, 31, dst.data_ptr(), 0), bm.synchronize)
        r["device_copy"] = timed(lambda: dst.copy_(src), torch.cuda.synchronize)
        r["prefilter_vs_copy"] = round(r["prefilter"]["GBps"] / r["device_copy"]["GBps"], 3)
        out.append(r)
        del src, dst
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
