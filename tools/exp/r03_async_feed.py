#!/usr/bin/env python3
"""Why did the asynchronous whole-batch feed (sbm_submit_dense / sbm_wait_oldest) run at 3.5 ms per 64 KITTI pairs in one
context and 1.7 ms in another?  Times the same loop before / after other transfer activity of the process."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import _pkg
pkg = _pkg.load()
from u96_slam_amd import synth
W, H, nd, B = 1242, 375, 128, 64
L, R = synth.make_batch(0, 16, W, H, nd)
L, R = np.concatenate([L] * 4), np.concatenate([R] * 4)
dev = torch.device("cuda", 0)
bm = pkg.StereoBM.create(nd, 15, device=0)
bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
sets = [(torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory(), torch.empty((B, H, W), dtype=torch.int16).pin_memory()) for _ in range(3)]
nps = [(a.numpy(), b.numpy(), c.numpy()) for a, b, c in sets]
dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
dD = torch.empty((B, H, W), dtype=torch.int16, device=dev)

def async_loop(n=20, depth=3):
    bm.synchronize()
    t = time.perf_counter()
    for i in range(n):
        bm.submit_host(*nps[i % 3])
        if i >= depth - 1:
            bm.wait_host()
    bm.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def sync_loop(n=10):
    bm.synchronize()
    t = time.perf_counter()
    for i in range(n):
        bm.compute(*nps[0])
    return (time.perf_counter() - t) / n * 1e3

print("async (first thing in the process)", round(async_loop(), 3), round(async_loop(), 3))
print("async depth 2", round(async_loop(depth=2), 3), "depth 1 (serial)", round(async_loop(depth=1), 3))
sets[0][2].copy_(dD, non_blocking=True); torch.cuda.synchronize()
print("async after one torch D2H", round(async_loop(), 3))
dL.copy_(sets[0][0], non_blocking=True); torch.cuda.synchronize()
print("async after one torch H2D", round(async_loop(), 3))
print("sync chunked", round(sync_loop(), 3))
print("async after sync path", round(async_loop(), 3), round(async_loop(), 3))
for _ in range(5):
    bm.launch_raw(B, dL.data_ptr(), dR.data_ptr(), W, H, dD.data_ptr())
bm.synchronize()
print("async after resident steps", round(async_loop(), 3))
ok = all(np.array_equal(nps[0][2], nps[k][2]) for k in (1, 2))
print("maps of the three buffer sets identical:", ok)

# when does a fresh handle's feed become fast?  per-step host times of one long loop on a NEW handle
bm2 = pkg.StereoBM.create(nd, 15, device=0)
bm2.setPreFilterCap(31); bm2.setTextureThreshold(10); bm2.setUniquenessRatio(10); bm2.setSpeckleWindowSize(50); bm2.setSpeckleRange(32); bm2.setDisp12MaxDiff(1)
ts = []
t = time.perf_counter()
for i in range(60):
    bm2.submit_host(*nps[i % 3])
    if i >= 2:
        bm2.wait_host()
    t2 = time.perf_counter(); ts.append((t2 - t) * 1e3); t = t2
bm2.synchronize()
print("new handle, per-step ms:", " ".join(f"{x:.2f}" for x in ts))
