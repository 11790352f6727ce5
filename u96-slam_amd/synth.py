"""Deterministic synthetic rectified stereo frames (SURVEY.md section 8d / BASELINE.md section 2).

pair i: rng = default_rng(0x5B4D0000 + i); texture T (H, W + 2 nd) uint8 smoothed by a 3x3 integer box mean;
banded ground-truth disparity D(y) = 8 + (37 * (y // 32)) mod (nd - 16);
L[y,x] = T[y, x + nd];  R[y,x] = clip(T[y, x + nd + D(y)] + n, 0, 255), n in {-2..2} per pixel  (so L(x) ~ R(x - D)).
"""
import numpy as np

SEED0 = 0x5B4D0000


def box3(t):
    """3x3 box mean with integer floor and edge replication."""
    p = np.pad(t.astype(np.uint16), 1, mode="edge")
    s = np.zeros(t.shape, np.uint16)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + t.shape[0], dx:dx + t.shape[1]]
    return (s // 9).astype(np.uint8)


def band_disparity(height, nd):
    y = np.arange(height)
    return 8 + (37 * (y // 32)) % max(nd - 16, 1)


def make_pair(index, width, height, nd):
    rng = np.random.default_rng(SEED0 + index)
    T = box3(rng.integers(0, 256, (height, width + 2 * nd), dtype=np.uint8))
    D = band_disparity(height, nd)
    cols = np.arange(width)
    L = T[:, nd:nd + width]
    idx = cols[None, :] + nd + D[:, None]
    R = np.take_along_axis(T, idx, axis=1).astype(np.int16) + rng.integers(-2, 3, (height, width), dtype=np.int16)
    return np.ascontiguousarray(L), np.clip(R, 0, 255).astype(np.uint8)


def make_batch(first_index, n, width, height, nd):
    Ls = np.empty((n, height, width), np.uint8)
    Rs = np.empty((n, height, width), np.uint8)
    for i in range(n):
        Ls[i], Rs[i] = make_pair(first_index + i, width, height, nd)
    return Ls, Rs
