#!/usr/bin/env python3
"""Experiment: does running two handles (two streams) on half batches overlap the post-filter of one with the SAD of the other?"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import _pkg
_pkg.load()
import torch
import numpy as np
from u96_slam_amd import StereoBM, synth

W, H, nd, w = 1242, 375, 128, 15
B = 64
L, R = synth.make_batch(0, 16, W, H, nd)
idx = [i % 16 for i in range(B)]
dev = torch.device("cuda:0")
dL = torch.from_numpy(np.ascontiguousarray(L[idx])).to(dev)
dR = torch.from_numpy(np.ascontiguousarray(R[idx])).to(dev)
dD = torch.empty((B, H, W), dtype=torch.int16, device=dev)

def mk():
    bm = StereoBM.create(nd, w)
    bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10)
    bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
    return bm

def run(nh, steps=30):
    hs = [mk() for _ in range(nh)]
    per = B // nh
    def step():
        for k, bm in enumerate(hs):
            o = k * per
            bm.launch_raw(per, dL[o:o+per].data_ptr(), dR[o:o+per].data_ptr(), W, H, dD[o:o+per].data_ptr())
    for _ in range(3): step()
    for bm in hs: bm.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    for bm in hs: bm.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    return dt

ref = None
for nh in (1, 2, 4, 8):
    dt = run(nh)
    out = dD.cpu().numpy().copy()
    if ref is None: ref = out
    print(f"handles={nh}: {dt:.3f} ms per {B} pairs, same={np.array_equal(ref, out)}", flush=True)
