#!/usr/bin/env python3
"""round 4 experiment: does running the post-filters of one half of the batch under the SAD kernel of the other half pay?
Two engine handles (each has its own stream), half the KITTI batch each, submitted back to back without synchronising, against
one handle with the whole batch. Steady-state wall time per 64 pairs."""
import sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import _pkg
pkg = _pkg.load()

def make(nd, wsz):
    bm = pkg.StereoBM.create(nd, wsz)
    bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32)
    bm.setDisp12MaxDiff(1)
    return bm

def run(shape, nd, wsz, n, parts, steps=200):
    h, w = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    L = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda", generator=g)
    R = torch.roll(L, -7, dims=2).contiguous()
    bms = [make(nd, wsz) for _ in range(parts)]
    m = n // parts
    Ls = [L[i * m:(i + 1) * m].contiguous() for i in range(parts)]; Rs = [R[i * m:(i + 1) * m].contiguous() for i in range(parts)]
    outs = [torch.empty((m, h, w), dtype=torch.int16, device="cuda") for _ in range(parts)]
    def step():
        for i in range(parts):
            bms[i].compute_device(Ls[i], Rs[i], outs[i], sync=False)
    for _ in range(20): step()
    for b in bms: b.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    for b in bms: b.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    return dt, torch.cat(outs)

for name, shape, nd, wsz, n in [("kitti", (375, 1242), 128, 15, 64), ("ref640", (480, 640), 64, 21, 64), ("fhd", (1080, 1920), 256, 21, 16)]:
    ref = None
    for parts in (1, 2, 4):
        dt, out = run(shape, nd, wsz, n, parts)
        if ref is None: ref = out
        print(name, "handles", parts, "ms per batch %.4f" % dt, "equal", bool(torch.equal(ref, out)), flush=True)
