"""round 6: repeated whole-map checks of the speckle filter's band walk (a race shows up as a non-deterministic mismatch):
the reference's 640x480 pair (windows 9 and 21), KITTI-shaped batches and a noisy map, speckle windows 50 / 2048 / 5000,
REPS launches each, every map against the oracle. usage: SBM_SPECKLE_BAND=<0|2|4> python3 tools/exp/r06_spk_reps.py [reps]"""
import os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import _pkg, sbm_oracle
pkg = _pkg.load()
from u96_slam_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = np.load(ROOT / "tests/golden/ref_pair_640x480.npz")
rng = np.random.default_rng(5)
cases = [("ref w9", g["rect_l"][None], g["rect_r"][None], 64, 9, 32), ("ref w21", g["rect_l"][None], g["rect_r"][None], 64, 21, 32)]
L, R = synth.make_batch(3, 6, 1242, 375, 128); cases.append(("kitti x6", L, R, 128, 15, 32))
L, R = synth.make_batch(4, 70, 640, 480, 64); cases.append(("vga x70", L, R, 64, 21, 16))
cases.append(("noise", rng.integers(0, 256, (3, 200, 900), dtype=np.uint8), rng.integers(0, 256, (3, 200, 900), dtype=np.uint8), 64, 5, 2))
bad = 0
for name, L, R, nd, w, rng_ in cases:
    for spw in (50, 2048, 5000):
        p = sbm_oracle.make_params(nd, w, 31, 0, 10 if name != "noise" else 0, 10 if name != "noise" else 0, spw, rng_, 1 if name != "noise" else -1)
        nref = min(len(L), 6)
        ref = sbm_oracle.compute_batch(p, L[:nref], R[:nref])
        bm = pkg.StereoBM.create(nd, w)
        bm.setTextureThreshold(p.texture_threshold); bm.setUniquenessRatio(p.uniqueness_ratio); bm.setDisp12MaxDiff(p.disp12_max_diff)
        bm.setSpeckleWindowSize(spw); bm.setSpeckleRange(rng_)
        Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        mm = []
        for _ in range(reps):
            got = bm.compute_device(Ld, Rd).cpu().numpy()
            mm.append(int((got[:nref] != ref).sum()) + int((got[nref:] != got[nref:]).sum()))
        bad += sum(mm)
        print(f"band={os.environ.get('SBM_SPECKLE_BAND', 'auto')} {name} spw={spw} mismatches {mm}", flush=True)
print("TOTAL", bad)
sys.exit(1 if bad else 0)
