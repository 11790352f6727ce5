#!/usr/bin/env python3
"""Soak test on the GPU: many seeded random configurations (sizes, windows 5..29, disparity counts, every post-filter
combination, batches; since round 5 windows up to 41 and up to 528 disparities: the 3 / 4-wavefront layouts and the sliding-sum
fallback kernel; since round 6 a random band height and segment count of the speckle filter's band walk and speckle windows
beyond its 2048-pixel limit) against the oracle, stage by stage; every configuration is also run twice for determinism.
usage: python tools/soak.py [--iters 400] [--seed 1]   -> prints a JSON summary, exit code 1 on any mismatch"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "oracle"))


def make_bm(pkg, kw):
    bm = pkg.StereoBM.create(kw.get("num_disparities", 64), kw.get("block_size", 21))
    setters = dict(prefilter_cap=bm.setPreFilterCap, min_disparity=bm.setMinDisparity, texture_threshold=bm.setTextureThreshold,
                   uniqueness_ratio=bm.setUniquenessRatio, speckle_window_size=bm.setSpeckleWindowSize,
                   speckle_range=bm.setSpeckleRange, disp12_max_diff=bm.setDisp12MaxDiff, prefilter_type=bm.setPreFilterType,
                   prefilter_size=bm.setPreFilterSize)
    for k, v in kw.items():
        if k in setters:
            setters[k](v)
    return bm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="frame-sized images and batches up to 48 pairs (the band walk's 4-row "
                    "bands, multi-iteration LR rows, packed border segments); a few hundred iterations take minutes")
    args = ap.parse_args()
    import numpy as np
    import _pkg

    pkg = _pkg.load()
    import sbm_oracle as oracle
    from test_gpu_parity import assert_stages_equal, rand_pair, run_engine

    rng = np.random.default_rng(args.seed)
    bad = []
    t0 = time.time()
    for it in range(args.iters):
        wsz = int(rng.choice([5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 41]))
        nd = int(rng.choice([16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 192, 208, 256, 272, 320, 384, 400, 512, 528]))
        mind = int(rng.choice([0, 0, 0, -16, 5, -nd // 2, 17]))
        n = int(rng.choice([1, 1, 2, 3, 5]))
        h = int(rng.integers(wsz + 8, wsz + 90))
        lo = max(nd + abs(mind), 0) + 2 * wsz + 8
        w = int(rng.integers(lo, lo + 300))
        if args.big:
            n = int(rng.choice([1, 4, 12, 24, 48]))
            h = int(rng.integers(180, 520))
            w = int(rng.integers(max(lo, 600), 1400))
        kw = dict(num_disparities=nd, block_size=wsz, min_disparity=mind, prefilter_cap=int(rng.choice([31, 31, 15, 63, 1, 40])),
                  texture_threshold=int(rng.choice([0, 10, 10, 200, 1000])), uniqueness_ratio=int(rng.choice([0, 5, 10, 15, 40, 90])),
                  disp12_max_diff=int(rng.choice([-1, 0, 1, 1, 3])))
        if rng.random() < 0.15:
            kw.update(prefilter_type=0, prefilter_size=int(rng.choice([5, 9, 9, 15, 31, 63])))
        if rng.random() < 0.6:
            kw.update(speckle_window_size=int(rng.choice([1, 10, 50, 200, 1000, 2048, 3000])), speckle_range=int(rng.choice([0, 4, 16, 32, 100])))
        # round 6: the speckle filter's band walk in every shape (band height x column segments per band; unset = automatic)
        for var, choices in (("SBM_SPECKLE_BAND", ["", "", "2", "4"]), ("SBM_SPECKLE_SEG", ["", "", "1", "2", "4"])):
            v = str(rng.choice(choices))
            os.environ.pop(var, None)
            if v:
                os.environ[var] = v
        pairs = [rand_pair(rng, h, w, shift=int(rng.integers(0, 14)), noise=int(rng.integers(0, 8))) for _ in range(n)]
        L = np.stack([p[0] for p in pairs]); R = np.stack([p[1] for p in pairs])
        if rng.random() < 0.3:
            L = (L // 64 * 64).astype(np.uint8); R = (R // 64 * 64).astype(np.uint8)
        if n == 1 and rng.random() < 0.5:
            L, R = L[0], R[0]
        try:
            if args.big and n >= 16:
                # the host entry point pipelines batches of >= 16 pairs in chunks (the per-stage planes then belong to the
                # last chunk only): check its final maps, then run the whole batch as ONE device call and check every stage
                eng, ref = run_engine(pkg, oracle, kw, L, R, stages=False)
                assert np.array_equal(eng["disp"].reshape(ref["disp"].shape), ref["disp"]), "final disparity differs (host, pipelined)"
                import torch

                bm = make_bm(pkg, kw)
                dd = bm.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
                dev = dict(disp=dd.cpu().numpy().reshape(n, h, w), pf_l=bm.debug_fetch(0, n, h, w), pf_r=bm.debug_fetch(1, n, h, w),
                           pre_lr=bm.debug_fetch(3, n, h, w))
                if kw.get("disp12_max_diff", -1) >= 0:
                    dev["cost"] = bm.debug_fetch(2, n, h, w)
                assert_stages_equal(dev, ref, kw)
            else:
                eng, ref = run_engine(pkg, oracle, kw, L, R)
                assert_stages_equal(eng, ref, kw)
                eng2, _ = run_engine(pkg, oracle, kw, L, R, stages=False)
                assert np.array_equal(eng2["disp"], eng["disp"]), "second run differs"
        except AssertionError as e:
            bad.append({"iteration": it, "shape": [n, h, w], "params": kw, "error": str(e)[:200]})
    # ---- the reference's own PL blocks (FPGA-flavour matcher, GFTT map): random sizes / windows / phases / filter settings
    import torch

    bmf = pkg.StereoBM.create(64, 21)
    nf = 0 if args.big else max(10, args.iters // 4)
    for it in range(nf):
        wsz = int(rng.choice([3, 5, 9, 15, 21, 27, 31]))
        nd = int(rng.choice([32, 64, 96, 128, 192, 256]))
        hw = wsz // 2
        if (nd + hw + 1) % 32 == 0:
            continue
        n = int(rng.choice([1, 2, 3]))
        h = int(rng.integers(wsz + 2, min(511, wsz + 140)))
        w = int(rng.integers(nd + wsz + 3, min(1023, nd + wsz + 400)))
        amp = int(rng.choice([64, 64, 2]))
        xr = (rng.integers(0, amp, (n, h, w)) * (63 if amp == 2 else 1)).astype(np.uint8)
        xl = np.stack([np.roll(xr[i], int(rng.integers(0, nd)), axis=1) for i in range(n)])
        if amp != 2:
            xl = np.clip(xl.astype(int) + rng.integers(-5, 6, xl.shape), 0, 63).astype(np.uint8)
        uni = (int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 1024)))
        try:
            fp = pkg.fpga_params(w, h, wsz, nd, *uni)
            got = bmf.fpga_bm(torch.from_numpy(xl).cuda(), torch.from_numpy(xr).cuda(), fp).cpu().numpy()
            for i in range(n):
                ref = oracle.fpga_bm(xl[i], xr[i], wsz, nd, *uni)
                assert np.array_equal(got[i], ref), f"fpga pair {i}: {(got[i] != ref).sum()} pixels"
            img = rng.integers(0, 256, (n, h, w), dtype=np.uint8) if h >= 5 else None
            if img is not None:
                eig, mx = bmf.gftt_eig(torch.from_numpy(img).cuda())
                for i in range(n):
                    re_, rm = oracle.gftt_eig(img[i])
                    assert np.array_equal(eig[i].cpu().numpy(), re_.astype(np.int64)) and int(mx[i]) == rm, f"gftt image {i}"
        except AssertionError as e:
            bad.append({"iteration": f"pl{it}", "shape": [n, h, w], "params": dict(wsz=wsz, nd=nd, uni=uni, amp=amp), "error": str(e)[:200]})
    print(json.dumps({"iterations": args.iters, "pl_iterations": nf, "seed": args.seed, "mismatches": len(bad),
                      "seconds": round(time.time() - t0, 1), "first": bad[:3]}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
