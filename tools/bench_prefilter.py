#!/usr/bin/env python3
"""Prefilter kernel against the device's own copy rate, inside and beyond the 256 MB Infinity Cache.

For each working set (images in + planes out): stand-alone prefilter kernel (sbm_prefilter_device, cv flavour), a plain
device-to-device copy of the same bytes (torch copy_ = the runtime's copy kernel) and a hand-rolled 16-byte-per-lane copy
kernel are timed over back-to-back launches; algorithmic bytes = 1 B read + 1 B written per pixel.
usage: python tools/bench_prefilter.py [--reps 30]"""
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
    args = ap.parse_args()
    import torch
    import _pkg

    pkg = _pkg.load()
    bm = pkg.StereoBM.create(64, 21)
    L = bm._L
    out = []
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
