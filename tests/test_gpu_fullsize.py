"""GPU tests at BASELINE.json's full per-GPU sizes, through size-independent properties (the oracle cannot run whole
batches of these in seconds): batch independence (every pair of a batch equals the same pair computed alone), equal inputs
-> equal outputs, a row band checked bit-exactly against the oracle (SAD/WTA/LR are row-local up to a w/2 halo), output
range and never-valid border, and the speckle post-condition (no surviving component of <= speckleWindowSize pixels,
verified with an independent scipy labelling)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FULL = dict(prefilter_cap=31, texture_threshold=10, uniqueness_ratio=10, speckle_window_size=50, speckle_range=32,
            disp12_max_diff=1)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


def make_engine(pkg, nd, wsz, **kw):
    bm = pkg.StereoBM.create(nd, wsz)
    bm.setPreFilterCap(kw.get("prefilter_cap", 31))
    bm.setTextureThreshold(kw.get("texture_threshold", 10))
    bm.setUniquenessRatio(kw.get("uniqueness_ratio", 10))
    bm.setSpeckleWindowSize(kw.get("speckle_window_size", 0))
    bm.setSpeckleRange(kw.get("speckle_range", 0))
    bm.setDisp12MaxDiff(kw.get("disp12_max_diff", -1))
    return bm


def device_batch(torch, L, R, reps):
    """n = len(L) * reps pairs on the device: the unique pairs repeated (pair i = unique pair i % len(L))."""
    dev = torch.device("cuda:0")
    dL = torch.from_numpy(L).to(dev).repeat(reps, 1, 1).contiguous()
    dR = torch.from_numpy(R).to(dev).repeat(reps, 1, 1).contiguous()
    return dL, dR


def crc_rows(a):
    """checksum of per-pair checksums"""
    per = [zlib.crc32(np.ascontiguousarray(x).tobytes()) for x in a]
    return per, zlib.crc32(np.asarray(per, np.uint32).tobytes())


def small_components_left(disp, win, rng_diff, filtered):
    """number of surviving 4-connected components with <= win pixels (edges: both valid and |delta| <= rng_diff)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    h, w = disp.shape
    d = disp.astype(np.int32)
    valid = d != filtered
    idx = np.arange(h * w).reshape(h, w)
    eh = valid[:, :-1] & valid[:, 1:] & (np.abs(d[:, :-1] - d[:, 1:]) <= rng_diff)
    ev = valid[:-1, :] & valid[1:, :] & (np.abs(d[:-1, :] - d[1:, :]) <= rng_diff)
    src = np.concatenate([idx[:, :-1][eh], idx[:-1, :][ev]])
    dst = np.concatenate([idx[:, 1:][eh], idx[1:, :][ev]])
    g = coo_matrix((np.ones(src.size, np.int8), (src, dst)), shape=(h * w, h * w))
    _, lab = connected_components(g, directed=False)
    sizes = np.bincount(lab[valid.ravel()], minlength=1)
    return int(((sizes > 0) & (sizes <= win)).sum())


def small_components_left_interior(disp, win, rng_diff, filtered):
    """like small_components_left, ignoring components that touch the border of the array (a crop cuts them)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    h, w = disp.shape
    d = disp.astype(np.int32)
    valid = d != filtered
    idx = np.arange(h * w).reshape(h, w)
    eh = valid[:, :-1] & valid[:, 1:] & (np.abs(d[:, :-1] - d[:, 1:]) <= rng_diff)
    ev = valid[:-1, :] & valid[1:, :] & (np.abs(d[:-1, :] - d[1:, :]) <= rng_diff)
    src = np.concatenate([idx[:, :-1][eh], idx[:-1, :][ev]])
    dst = np.concatenate([idx[:, 1:][eh], idx[1:, :][ev]])
    g = coo_matrix((np.ones(src.size, np.int8), (src, dst)), shape=(h * w, h * w))
    _, lab = connected_components(g, directed=False)
    lab = lab.reshape(h, w)
    sizes = np.bincount(lab[valid], minlength=lab.max() + 1)
    border = np.zeros_like(valid)
    border[0, :] = border[-1, :] = border[:, 0] = border[:, -1] = True
    touching = np.unique(lab[valid & border])
    small = (sizes > 0) & (sizes <= win)
    small[touching] = False
    return int(small.sum())


def check_common(out, W, H, nd, wsz, filtered=-16):
    # never-valid frame (cv getValidDisparityROI): left nd-1+w/2 columns, right w/2 columns, w/2 rows top and bottom
    h2 = wsz // 2
    assert (out[:, :h2, :] == filtered).all() and (out[:, H - h2:, :] == filtered).all()
    assert (out[:, :, : nd - 1 + h2] == filtered).all() and (out[:, :, W - h2:] == filtered).all()
    v = out[out != filtered]
    assert v.size > 0 and v.min() >= 0 and v.max() <= (nd - 1) * 16 + 15


def band_vs_oracle(pkg, oracle, L, R, nd, wsz, y0, rows, got_full):
    """SAD/WTA/uniqueness/LR of rows [y0, y0+rows) depend only on input rows [y0-w/2-1, y0+rows+w/2+1): compare the
    engine's full-frame result (speckle off) with the oracle run on that band alone."""
    h2 = wsz // 2
    a, b = y0 - h2 - 1, y0 + rows + h2 + 1
    p = oracle.make_params(num_disparities=nd, block_size=wsz, prefilter_cap=31, texture_threshold=10,
                           uniqueness_ratio=10, disp12_max_diff=1)
    ref = oracle.compute(p, np.ascontiguousarray(L[a:b]), np.ascontiguousarray(R[a:b]))
    sub = ref[h2 + 1: h2 + 1 + rows]
    assert np.array_equal(got_full[y0:y0 + rows], sub), "row band differs from the oracle"


@pytest.mark.parametrize("name,W,H,nd,wsz,n,uniq", [
    ("configs[2]: 1920x1080 nd256, 64 pairs", 1920, 1080, 256, 21, 64, 2),
    ("configs[3] per-GPU share: 1242x375 nd128, 64 pairs", 1242, 375, 128, 15, 64, 4),
    ("configs[4] per-GPU share: 3840x2160 nd256, 32 pairs, every post-filter", 3840, 2160, 256, 21, 32, 2),
])
def test_full_size_batch_properties(torch_cuda, pkg, oracle, name, W, H, nd, wsz, n, uniq):
    torch = torch_cuda
    from u96_slam_amd import synth

    L, R = synth.make_batch(100, uniq, W, H, nd)
    # the last unique pair gets heavy sensor noise so that mismatches, LR rejections and speckles exist at this size
    rng = np.random.default_rng(4242)
    R[uniq - 1] = np.clip(R[uniq - 1].astype(np.int16) + rng.integers(-48, 49, (H, W), dtype=np.int16), 0, 255).astype(np.uint8)
    dL, dR = device_batch(torch, L, R, n // uniq)
    bm = make_engine(pkg, nd, wsz, **FULL)
    out = bm.compute_device(dL, dR).cpu().numpy()
    assert out.shape == (n, H, W) and out.dtype == np.int16

    # equal inputs -> equal outputs wherever they sit in the batch; checksum of checksums is reproducible
    per, total = crc_rows(out)
    for i in range(n):
        assert per[i] == per[i % uniq], f"pair {i} differs from its duplicate {i % uniq}"
    out2 = bm.compute_device(dL, dR).cpu().numpy()
    assert crc_rows(out2)[1] == total

    # batch independence: a pair computed alone (fresh handle, batch of 1) equals the same pair inside the batch
    solo = make_engine(pkg, nd, wsz, **FULL)
    for u in range(uniq):
        alone = solo.compute_device(dL[u:u + 1], dR[u:u + 1]).cpu().numpy()[0]
        assert np.array_equal(alone, out[u])

    check_common(out[:uniq], W, H, nd, wsz)
    assert (out[:uniq] >= 0).mean() > 0.3
    # speckle post-condition on the noisy frame
    assert small_components_left(out[uniq - 1], 50, 32, -16) == 0

    # a 48-row band against the oracle, speckle off (the only image-global stage)
    nosp = dict(FULL, speckle_window_size=0, speckle_range=0)
    bm2 = make_engine(pkg, nd, wsz, **nosp)
    got = bm2.compute_device(dL[:uniq], dR[:uniq]).cpu().numpy()
    for u in (0, uniq - 1):
        band_vs_oracle(pkg, oracle, L[u], R[u], nd, wsz, H // 2 - 10, 48, got[u])
    # speckle filtering only ever removes pixels, and on the noisy frame it has work to do
    changed = out[:uniq] != got
    assert (out[:uniq][changed] == -16).all()
    assert small_components_left(got[uniq - 1], 50, 32, -16) > 0 and changed[uniq - 1].sum() > 0


def test_shift_covariance_full_hd(torch_cuda, pkg):
    """Translating the right image by k whole pixels moves every SAD curve by k indices: away from the frame and from
    the ends of the disparity range the map changes by exactly 16*k (sub-pixel terms included)."""
    torch = torch_cuda
    from u96_slam_amd import synth

    W, H, nd, wsz, k = 1920, 1080, 256, 21, 5
    rng = np.random.default_rng(77)
    T = synth.box3(rng.integers(0, 256, (H, W + 128), dtype=np.uint8))
    L = np.ascontiguousarray(T[:, 32:32 + W])
    R0 = np.ascontiguousarray(T[:, 32 + 40:32 + 40 + W])          # disparity 40
    R1 = np.ascontiguousarray(T[:, 32 + 40 + k:32 + 40 + k + W])  # disparity 40 + k
    bm = make_engine(pkg, nd, wsz, texture_threshold=10, uniqueness_ratio=10)
    dev = torch.device("cuda:0")
    dL = torch.from_numpy(np.stack([L, L])).to(dev)
    dR = torch.from_numpy(np.stack([R0, R1])).to(dev)
    out = bm.compute_device(dL, dR).cpu().numpy()
    a, b = out[0], out[1]
    inner = np.zeros((H, W), bool)
    inner[wsz:H - wsz, nd + wsz: W - wsz - 64] = True
    both = inner & (a >= 0) & (b >= 0)
    assert both.mean() > 0.5
    assert np.array_equal(b[both], a[both] + 16 * k)
    assert (np.abs(a[both].astype(int) - 40 * 16) <= 8).mean() > 0.99   # whole-pixel truth, sub-pixel term within half a pixel


def test_very_large_frame(torch_cuda, pkg, oracle):
    """One 8192x2048 frame (wider than the 16-bit-key LR kernel's 4096 columns, 16.8 Mpx of labels): row bands against the
    oracle, frame and range properties, speckle post-condition on a crop."""
    torch = torch_cuda
    from u96_slam_amd import synth

    W, H, nd, wsz = 8192, 2048, 256, 21
    L, R = synth.make_pair(55, W, H, nd)
    rng = np.random.default_rng(9)
    R[H // 2:] = np.clip(R[H // 2:].astype(np.int16) + rng.integers(-48, 49, (H - H // 2, W), dtype=np.int16), 0, 255).astype(np.uint8)
    dL, dR = torch.from_numpy(L[None]).cuda(), torch.from_numpy(R[None]).cuda()
    out = make_engine(pkg, nd, wsz, **FULL).compute_device(dL, dR).cpu().numpy()
    check_common(out, W, H, nd, wsz)
    nosp = dict(FULL, speckle_window_size=0, speckle_range=0)
    got = make_engine(pkg, nd, wsz, **nosp).compute_device(dL, dR).cpu().numpy()[0]
    for y0 in (wsz, H // 2 - 16, H - 2 * wsz - 40):
        band_vs_oracle(pkg, oracle, L, R, nd, wsz, y0, 32, got)
    changed = out[0] != got
    assert (out[0][changed] == -16).all() and changed.sum() > 0
    # components cut by a crop border look smaller than they are, so check the crop of the FILTERED map only for components
    # that do not touch the crop border: none of them may be small
    crop = out[0][H // 2 + 100:H // 2 + 400, 1000:3000]
    assert small_components_left_interior(crop, 50, 32, -16) == 0
