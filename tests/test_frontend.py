"""Producers in front of the path (SURVEY.md 8f rank 2 + the prefilter half of rank 3): rectification map
(fpga.c:303-366), bilinear resampling (rect_intp.v:285-404), stand-alone x-Sobel in both flavours.

CPU part: the oracle restatement against an independent big-integer Python restatement and against algebraic
properties. GPU part: the HIP kernels, through the C-ABI, bit-exact against the oracle -- and the RTL-flavour prefilter
directly against the reference-produced golden vectors data/ref_xsbl_{l,r} (pinned)."""
import numpy as np
import pytest

# calibration data set "220426" of the reference firmware (fpga.c:190-226): values only, quoted as test vectors
CAM_L = dict(f=(40419817, 40382910), c=(320, 240), f2inv=(6338213, 6338213), c2_f2=(4984405, 5932596),
             rot=[[16598538, -120818, 2439034], [137992, 16776300, -108069], [-2438123, 126979, 16598626]])
CAM_R = dict(f=(39609530, 39627967), c=(320, 240), f2inv=(6338213, 6338213), c2_f2=(4984405, 5932596),
             rot=[[16569087, -69780, 2633522], [51223, 16776692, 122251], [-2633948, -112694, 16568783]])


def py_rect_point(cam, xd, yd):
    """Independent restatement with Python integers (arbitrary precision; >> is an arithmetic shift)."""
    xn = ((xd * cam["f2inv"][0]) >> 8) - cam["c2_f2"][0]
    yn = ((yd * cam["f2inv"][1]) >> 8) - cam["c2_f2"][1]
    r = cam["rot"]
    lx = ((r[0][0] * xn) >> 24) + ((r[1][0] * yn) >> 24) + r[2][0]
    ly = ((r[0][1] * xn) >> 24) + ((r[1][1] * yn) >> 24) + r[2][1]
    lw = ((r[0][2] * xn) >> 24) + ((r[1][2] * yn) >> 24) + r[2][2]
    den = lw % (1 << 64)                       # conversion to unsigned long long
    inv = ((1 << 48) // den) if den else 0
    out = []
    for num, f, c in ((lx, cam["f"][0], cam["c"][0]), (ly, cam["f"][1], cam["c"][1])):
        v = ((((num * inv) >> 24) * f) >> 34) + (c << 6)
        v = (v + 1) >> 1
        v &= 0xFFFF
        out.append(v - 0x10000 if v & 0x8000 else v)
    return out


def scaled_cam(cam, W, H):
    """The same rig on a W x H sensor (focal lengths and centres scaled): gives maps for non-VGA sizes."""
    sx, sy = W / 640.0, H / 480.0
    c = dict(cam)
    c["f"] = (int(cam["f"][0] * sx), int(cam["f"][1] * sy))
    c["c"] = (int(320 * sx), int(240 * sy))
    c["f2inv"] = (int(cam["f2inv"][0] / sx), int(cam["f2inv"][1] / sy))
    return c


def py_remap_point(src, mx, my):
    h, w = src.shape
    xi, yi, xf, yf = mx >> 5, my >> 5, mx & 31, my & 31
    t = lambda x, y: int(src[y, x]) if (0 <= x < w and 0 <= y < h) else 0
    acc = (t(xi, yi) * (32 - xf) * (32 - yf) + t(xi + 1, yi) * xf * (32 - yf) + t(xi, yi + 1) * (32 - xf) * yf
           + t(xi + 1, yi + 1) * xf * yf)
    return ((acc >> 9) + 1) >> 1


# ------------------------------------------------------------------------------------------------ CPU (oracle) -----
def test_oracle_rect_map_matches_python_integers(oracle):
    rng = np.random.default_rng(5)
    for cam in (CAM_L, CAM_R, scaled_cam(CAM_L, 1242, 375)):
        W, H = (640, 480) if cam["c"][0] == 320 else (1242, 375)
        m = oracle.rect_map(oracle.make_rect_cam(**cam), W, H)
        pts = [(0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)] + [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(300)]
        for x, y in pts:
            assert list(m[y, x]) == py_rect_point(cam, x, y), (x, y)


def test_oracle_rect_map_firmware_calibration_is_sane(oracle):
    """With the firmware's own calibration every destination pixel samples inside the 640x480 source, coordinates
    increase along rows/columns, and the correction stays a few tens of pixels."""
    for cam in (CAM_L, CAM_R):
        m = oracle.rect_map(oracle.make_rect_cam(**cam), 640, 480).astype(np.int32)
        x, y = m[..., 0] / 32.0, m[..., 1] / 32.0
        assert x.min() >= 0 and x.max() <= 639 and y.min() >= 0 and y.max() <= 479
        assert (np.diff(m[..., 0], axis=1) > 0).all() and (np.diff(m[..., 1], axis=0) > 0).all()
        gx, gy = np.meshgrid(np.arange(640), np.arange(480))
        assert np.abs(x - gx).max() < 64 and np.abs(y - gy).max() < 64


def test_oracle_rect_remap_properties(oracle):
    rng = np.random.default_rng(6)
    h, w = 37, 53
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    gx, gy = np.meshgrid(np.arange(w), np.arange(h))
    ident = np.stack([gx * 32, gy * 32], -1).astype(np.int16)
    assert np.array_equal(oracle.rect_remap(src, ident), src)            # zero fractions reproduce the source
    half = ident.copy(); half[..., 0] += 16
    got = oracle.rect_remap(src, half)[:, :-1].astype(int)
    s = src.astype(int)
    assert np.array_equal(got, ((((s[:, :-1] + s[:, 1:]) * 512) >> 9) + 1) >> 1)   # mean of two taps, round half up
    const = np.full((h, w), 200, np.uint8)
    frac = ident.copy(); frac[..., 0] += 7; frac[..., 1] += 21
    assert (oracle.rect_remap(const, frac)[:-1, :-1] == 200).all()       # weights sum to 1024
    # arbitrary maps, including taps outside the frame (read as 0)
    wild = rng.integers(-200, 32 * 60, (h, w, 2)).astype(np.int16)
    out = oracle.rect_remap(src, wild)
    for _ in range(400):
        y, x = int(rng.integers(h)), int(rng.integers(w))
        assert out[y, x] == py_remap_point(src, int(wild[y, x, 0]), int(wild[y, x, 1]))


# ------------------------------------------------------------------------------------------------ GPU --------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(640, 480), (1242, 375), (333, 77), (1920, 1080)])
def test_gpu_rect_map_bit_exact(torch_cuda, pkg, oracle, W, H):
    bm = pkg.StereoBM.create(64, 21)
    for base in (CAM_L, CAM_R):
        cam = base if (W, H) == (640, 480) else scaled_cam(base, W, H)
        got = bm.rect_map(pkg.make_rect_cam(**cam), W, H).cpu().numpy()
        ref = oracle.rect_map(oracle.make_rect_cam(**cam), W, H)
        assert np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,n", [(640, 480, 3), (1242, 375, 2), (333, 77, 5), (61, 35, 1)])
def test_gpu_rect_remap_bit_exact(torch_cuda, pkg, oracle, W, H, n):
    torch = torch_cuda
    rng = np.random.default_rng(W * 7 + H)
    src = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    bm = pkg.StereoBM.create(64, 21)
    cam = CAM_L if (W, H) == (640, 480) else scaled_cam(CAM_R, W, H)
    maps = [oracle.rect_map(oracle.make_rect_cam(**cam), W, H),
            rng.integers(-300, 32 * (max(W, H) + 8), (H, W, 2)).astype(np.int16)]   # incl. taps outside the frame
    for m in maps:
        got = bm.rect_remap(torch.from_numpy(src).cuda(), torch.from_numpy(m).cuda()).cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], oracle.rect_remap(src[i], m)), f"image {i}"
    # single-image form and a map built on the device
    dmap = bm.rect_map(pkg.make_rect_cam(**cam), W, H)
    one = bm.rect_remap(torch.from_numpy(src[0]).cuda(), dmap).cpu().numpy()
    assert np.array_equal(one, oracle.rect_remap(src[0], maps[0]))


@pytest.mark.gpu
def test_gpu_rtl_prefilter_reproduces_reference_golden(torch_cuda, pkg, oracle, golden):
    """PINNED: data/ref_xsbl_{l,r} is the reference RTL's x-Sobel of data/ref_rect_{l,r}."""
    torch = torch_cuda
    bm = pkg.StereoBM.create(64, 21)
    src = np.stack([golden["rect_l"], golden["rect_r"], golden["rect_l"]])   # odd count: exercises the tail launch
    got = bm.prefilter(torch.from_numpy(src).cuda(), flavour=pkg.PREFILTER_FLAVOUR_RTL).cpu().numpy()
    assert np.array_equal(got[0], golden["xsbl_l"]) and np.array_equal(got[1], golden["xsbl_r"])
    assert np.array_equal(got[2], golden["xsbl_l"])


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,n,cap", [(64, 40, 1, 31), (71, 37, 2, 15), (333, 77, 3, 63), (1242, 375, 4, 31), (17, 5, 2, 1)])
def test_gpu_standalone_prefilter_both_flavours(torch_cuda, pkg, oracle, W, H, n, cap):
    torch = torch_cuda
    rng = np.random.default_rng(W + H + n)
    src = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    bm = pkg.StereoBM.create(16, 5)
    d = torch.from_numpy(src).cuda()
    cv = bm.prefilter(d, flavour=pkg.PREFILTER_FLAVOUR_CV, cap=cap).cpu().numpy()
    rtl = bm.prefilter(d, flavour=pkg.PREFILTER_FLAVOUR_RTL).cpu().numpy()
    for i in range(n):
        assert np.array_equal(cv[i], oracle.prefilter_xsobel(src[i], cap))
        assert np.array_equal(rtl[i], oracle.prefilter_xsobel_fpga(src[i]))
    with pytest.raises(pkg.StereoBMError):
        bm.prefilter(d, flavour=7)
    with pytest.raises(pkg.StereoBMError):
        bm.prefilter(d, flavour=pkg.PREFILTER_FLAVOUR_CV, cap=64)


@pytest.mark.gpu
def test_gpu_raw_frames_to_disparity_chain(torch_cuda, pkg, oracle):
    """rectify -> block matching entirely on the device equals the same chain on the CPU oracle."""
    torch = torch_cuda
    from u96_slam_amd import synth

    W, H, nd = 640, 480, 64
    L, R = synth.make_pair(3, W, H, nd)
    bm = pkg.StereoBM.create(nd, 21)
    bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10)
    bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
    ml = bm.rect_map(pkg.make_rect_cam(**CAM_L), W, H)
    mr = bm.rect_map(pkg.make_rect_cam(**CAM_R), W, H)
    rl = bm.rect_remap(torch.from_numpy(L).cuda(), ml)
    rr = bm.rect_remap(torch.from_numpy(R).cuda(), mr)
    disp = bm.compute_device(rl, rr).cpu().numpy()
    ol = oracle.rect_remap(L, oracle.rect_map(oracle.make_rect_cam(**CAM_L), W, H))
    orr = oracle.rect_remap(R, oracle.rect_map(oracle.make_rect_cam(**CAM_R), W, H))
    p = oracle.make_params(num_disparities=nd, block_size=21, prefilter_cap=31, texture_threshold=10, uniqueness_ratio=10,
                           speckle_window_size=50, speckle_range=32, disp12_max_diff=1)
    assert np.array_equal(rl.cpu().numpy(), ol) and np.array_equal(rr.cpu().numpy(), orr)
    assert np.array_equal(disp, oracle.compute(p, ol, orr))
