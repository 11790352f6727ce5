import sys, time
sys.path.insert(0, '.')
import _pkg; pkg = _pkg.load()
import numpy as np
from u96_slam_amd import synth
L, R = synth.make_pair(0, 640, 480, 64)
def frame():
    bm = pkg.StereoBM.create(16, 9)
    bm.setPreFilterCap(31); bm.setBlockSize(21); bm.setMinDisparity(0); bm.setNumDisparities(64)
    bm.setTextureThreshold(10); bm.setUniquenessRatio(10); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
    d = bm.compute(L, R)
    del bm
    return d
for _ in range(3): frame()
t = time.perf_counter()
for _ in range(50): frame()
print("re-created per frame (main.cpp:201 pattern): %.3f ms per frame" % ((time.perf_counter() - t) / 50 * 1e3))
bm = pkg.StereoBM.create(64, 21)
bm.setTextureThreshold(10); bm.setUniquenessRatio(10); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
for _ in range(3): bm.compute(L, R)
t = time.perf_counter()
for _ in range(50): bm.compute(L, R)
print("handle kept: %.3f ms per frame" % ((time.perf_counter() - t) / 50 * 1e3))
