#!/usr/bin/env python3
"""round 4 scratch: a few small configurations of the engine against the oracle, stage by stage (pre-LR map + cost), with the
mismatch positions printed -- the quick loop behind the border-wavefront work. usage: python tools/exp/r04_quick.py"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import _pkg  # noqa: E402
import sbm_oracle  # noqa: E402

pkg = _pkg.load()
from u96_slam_amd import synth  # noqa: E402

import torch  # noqa: E402

cases = [(200, 64, 32, 9, 3), (320, 96, 64, 21, 5), (400, 80, 128, 15, 9), (640, 120, 256, 21, 2), (333, 77, 48, 11, 7), (300, 70, 96, 27, 17)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
bad = 0
for W, H, nd, w, n in cases:
    L, R = synth.make_batch(3, n, W, H, nd)
    bm = pkg.StereoBM.create(nd, w, device=0)
    bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(30); bm.setSpeckleRange(16)
    p = sbm_oracle.make_params(nd, w, 31, 0, 10, 10, 30, 16, 1)
    got = bm.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    pre = bm.debug_fetch(3, n, H, W)
    ref = sbm_oracle.compute_batch(p, L, R)
    st = sbm_oracle.stages(p, L[0], R[0]) if hasattr(sbm_oracle, "stages") else None
    ok = np.array_equal(got, ref)
    msg = f"{W}x{H} nd{nd} w{w} n{n}: final {'OK' if ok else 'MISMATCH ' + str(int((got != ref).sum()))} kernel {bm.last_kernel()}"
    if not ok:
        bad += 1
        ys, xs = np.nonzero((got != ref).any(axis=0))
        msg += f" cols {sorted(set(xs.tolist()))[:12]} rows {sorted(set(ys.tolist()))[:8]}"
    print(msg, flush=True)
sys.exit(1 if bad else 0)
