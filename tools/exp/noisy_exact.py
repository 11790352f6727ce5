import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import _pkg; pkg = _pkg.load()
import numpy as np, torch, sbm_oracle as oracle
from u96_slam_amd import synth
rng = np.random.default_rng(3)
for (W, H, nd, wsz, n) in ((1920, 1080, 256, 21, 2), (1242, 375, 128, 15, 8), (3840, 2160, 256, 21, 1)):
    L, R = synth.make_batch(200, n, W, H, nd)
    R = np.clip(R.astype(np.int16) + rng.integers(-40, 41, R.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    L[:, ::7, ::5] = 0   # speckle-provoking dropouts
    bm = pkg.StereoBM.create(nd, wsz)
    bm.setTextureThreshold(10); bm.setUniquenessRatio(10); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
    got = bm.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    p = oracle.make_params(num_disparities=nd, block_size=wsz, texture_threshold=10, uniqueness_ratio=10, speckle_window_size=50, speckle_range=32, disp12_max_diff=1)
    t = time.time(); ref = oracle.compute_batch(p, L, R); dt = time.time() - t
    print(W, H, nd, wsz, n, "equal:", np.array_equal(got, ref), "valid frac", (ref >= 0).mean().round(3), "oracle s", round(dt, 1), flush=True)
