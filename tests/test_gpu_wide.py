"""The fallback outside the fast kernel's envelope (sbm_sad_wide.hip: block sizes above 31, more than 512 disparities, sums
beyond 16 bits) against the oracle, stage by stage, and against the per-column kernel it replaced (SBM_WIDE=0; the switch is
read at every call). Bit-exact: integer path, tolerance 0."""
import pathlib
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


# (width, height, pairs, parameters): every case is outside the fast envelope
CASES = [
    (260, 90, 2, dict(num_disparities=64, block_size=33)),                                    # one chunk, one wavefront
    (400, 100, 1, dict(num_disparities=128, block_size=35, uniqueness_ratio=15)),             # two wavefronts
    (360, 120, 3, dict(num_disparities=96, block_size=45, texture_threshold=40)),             # masked lanes, big window
    (700, 70, 2, dict(num_disparities=528, block_size=15, uniqueness_ratio=10)),              # 9 chunks on 8 wavefronts
    (420, 60, 1, dict(num_disparities=272, block_size=33, uniqueness_ratio=10)),              # 5 chunks on 5 wavefronts
    (700, 60, 1, dict(num_disparities=512, block_size=37, uniqueness_ratio=10)),              # 8 wavefronts
    (1150, 40, 1, dict(num_disparities=1040, block_size=9, uniqueness_ratio=5)),              # 17 chunks: 3 per wavefront (<4>)
    (900, 36, 1, dict(num_disparities=768, block_size=11, uniqueness_ratio=0)),               # 12 chunks: 2 per wavefront (<2>)
    (2200, 30, 1, dict(num_disparities=2048, block_size=7, uniqueness_ratio=10)),             # the kernel's maximum
    (300, 80, 2, dict(num_disparities=64, block_size=27, prefilter_cap=63)),                  # sums beyond 16 bits
    (330, 90, 1, dict(num_disparities=48, block_size=33, min_disparity=-7, uniqueness_ratio=12)),
    (330, 90, 1, dict(num_disparities=80, block_size=35, min_disparity=11, texture_threshold=0, uniqueness_ratio=0)),
    (200, 150, 1, dict(num_disparities=32, block_size=101, uniqueness_ratio=10)),             # window wider than the tile's outputs
    (340, 110, 2, dict(num_disparities=64, block_size=31, roi1=(10, 3, 300, 100), roi2=(5, 5, 320, 100), uniqueness_ratio=10)),
    (310, 75, 1, dict(num_disparities=160, block_size=39, prefilter_type=0, prefilter_size=9, uniqueness_ratio=10)),
]


@pytest.mark.parametrize("lr", [-1, 1])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_wide_kernel_against_oracle_and_per_column_kernel(torch_cuda, pkg, oracle, case, lr, monkeypatch):
    from test_gpu_parity import assert_stages_equal, run_engine
    from u96_slam_amd import synth

    w, h, n, kw = CASES[case]
    kw = dict(dict(prefilter_cap=31, texture_threshold=10, uniqueness_ratio=0, speckle_window_size=20, speckle_range=8), **kw)
    kw["disp12_max_diff"] = lr
    L, R = synth.make_batch(40 + case, n, w, h, min(kw["num_disparities"], w // 3))
    got = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("SBM_WIDE", wide)
        eng, ref = run_engine(pkg, oracle, kw, L, R)
        got[wide] = eng
        assert_stages_equal(eng, ref, kw)
        assert np.array_equal(eng["disp"], ref["disp"]), f"SBM_WIDE={wide}: final disparity differs"
    assert np.array_equal(got["1"]["disp"], got["0"]["disp"])


def test_wide_kernel_is_the_one_that_runs(torch_cuda, pkg, monkeypatch):
    from u96_slam_amd import synth

    L, R = synth.make_batch(1, 1, 800, 80, 64)
    for wide, nd, wsz, want in (("1", 64, 33, "sad_wide_kernel"), ("0", 64, 33, "sad_generic_kernel"), ("1", 528, 15, "sad_wide_kernel"),
                                ("1", 320, 15, "sad_fast_kernel<128,3,"), ("1", 64, 21, "sad_fast_kernel<")):
        monkeypatch.setenv("SBM_WIDE", wide)
        bm = pkg.StereoBM.create(nd, wsz)
        bm.compute(L, R)
        assert bm.last_kernel().startswith(want), (wide, nd, wsz, bm.last_kernel())


def test_wide_kernel_frame_sized_properties(torch_cuda, pkg, monkeypatch):
    """Frame-sized launch beyond what the oracle checks in seconds: pairs are independent and the two fallback kernels agree
    (1280x720, 320 disparities, 35x35)."""
    from u96_slam_amd import synth

    L, R = synth.make_batch(9, 2, 1280, 720, 200)
    L4, R4 = np.concatenate([L, L]), np.concatenate([R, R])
    bm = pkg.StereoBM.create(320, 35)
    bm.setUniquenessRatio(10)
    bm.setDisp12MaxDiff(1)
    monkeypatch.setenv("SBM_WIDE", "1")
    a = bm.compute(L4, R4)
    assert np.array_equal(a[:2], a[2:])
    monkeypatch.setenv("SBM_WIDE", "0")
    b = bm.compute(L, R)
    assert np.array_equal(a[:2], b)
    assert (a >= 0).mean() > 0.2


# ---- 257 .. 512 disparities inside the interior kernel: three / four cooperating 128-disparity wavefronts, their clamped border
# ---- columns from the sliding-sum kernel (16-bit cost plane, pre-scaled planes)
ND512 = [
    (700, 64, 2, dict(num_disparities=272, block_size=15, uniqueness_ratio=10)),       # <128,3>, 16 disparities in the last wavefront
    (760, 70, 1, dict(num_disparities=384, block_size=21, uniqueness_ratio=15)),       # <128,3> full
    (800, 60, 3, dict(num_disparities=400, block_size=9, uniqueness_ratio=10)),        # <128,4>, 16 in the last wavefront
    (900, 66, 1, dict(num_disparities=512, block_size=27, uniqueness_ratio=10)),       # <128,4> full, 9-term window
    (820, 58, 2, dict(num_disparities=448, block_size=11, uniqueness_ratio=5)),        # 1-column sums
    (860, 72, 1, dict(num_disparities=320, block_size=19, uniqueness_ratio=0, texture_threshold=0)),
    (840, 64, 1, dict(num_disparities=496, block_size=5, min_disparity=-9, uniqueness_ratio=10)),
    (880, 90, 1, dict(num_disparities=336, block_size=25, min_disparity=13, uniqueness_ratio=10)),
    (1000, 50, 9, dict(num_disparities=512, block_size=15, uniqueness_ratio=10)),      # more pairs than XCDs
]


@pytest.mark.parametrize("lr", [-1, 1])
@pytest.mark.parametrize("case", range(len(ND512)))
def test_interior_kernel_up_to_512_disparities(torch_cuda, pkg, oracle, case, lr):
    from test_gpu_parity import assert_stages_equal, run_engine
    from u96_slam_amd import synth

    w, h, n, kw = ND512[case]
    kw = dict(dict(prefilter_cap=31, texture_threshold=10, speckle_window_size=20, speckle_range=8), **kw)
    kw["disp12_max_diff"] = lr
    L, R = synth.make_batch(70 + case, n, w, h, min(kw["num_disparities"], w // 3))
    eng, ref = run_engine(pkg, oracle, kw, L, R)
    assert_stages_equal(eng, ref, kw)
    assert np.array_equal(eng["disp"], ref["disp"])
    bm = pkg.StereoBM.create(kw["num_disparities"], kw["block_size"])
    bm.compute(L[:1], R[:1])
    nw = 3 if kw["num_disparities"] <= 384 else 4
    assert bm.last_kernel().startswith(f"sad_fast_kernel<128,{nw},"), bm.last_kernel()


def test_interior_kernel_512_frame_sized(torch_cuda, pkg, oracle):
    """One 2160p-wide band at 512 disparities, 21x21, every post-filter: whole map against the oracle (a few seconds of CPU)."""
    from u96_slam_amd import synth

    L, R = synth.make_batch(5, 1, 3840, 160, 400)
    kw = dict(num_disparities=512, block_size=21, prefilter_cap=31, texture_threshold=10, uniqueness_ratio=10,
              speckle_window_size=50, speckle_range=32, disp12_max_diff=1)
    bm = pkg.StereoBM.create(512, 21)
    for k, f in (("uniqueness_ratio", bm.setUniquenessRatio), ("disp12_max_diff", bm.setDisp12MaxDiff),
                 ("speckle_window_size", bm.setSpeckleWindowSize), ("speckle_range", bm.setSpeckleRange),
                 ("texture_threshold", bm.setTextureThreshold)):
        f(kw[k])
    got = bm.compute(L, R)
    p = oracle.make_params(**kw)
    ref = oracle.compute(p, L[0], R[0])
    assert np.array_equal(got[0], ref)
    assert bm.last_kernel().startswith("sad_fast_kernel<128,4,7,3,true")


# ---- windows 29 and 31 inside the interior kernel (1-column sums of 29 / 31 terms; sums still fit 16 bits at cap <= 34 / 31)
W2931 = [
    (400, 100, 2, dict(num_disparities=64, block_size=29, uniqueness_ratio=10)),
    (420, 110, 1, dict(num_disparities=128, block_size=31, uniqueness_ratio=15)),
    (360, 90, 3, dict(num_disparities=32, block_size=29, uniqueness_ratio=0, texture_threshold=0)),
    (700, 80, 1, dict(num_disparities=256, block_size=31, uniqueness_ratio=10)),
    (500, 96, 2, dict(num_disparities=96, block_size=31, uniqueness_ratio=10, min_disparity=-5)),
    (640, 120, 1, dict(num_disparities=192, block_size=29, uniqueness_ratio=10)),
    (900, 90, 1, dict(num_disparities=400, block_size=29, uniqueness_ratio=10)),
    (380, 100, 9, dict(num_disparities=48, block_size=31, uniqueness_ratio=5)),
    (640, 480, 1, dict(num_disparities=64, block_size=31, uniqueness_ratio=10)),
    (1242, 375, 2, dict(num_disparities=128, block_size=29, uniqueness_ratio=10, prefilter_cap=34)),
    (420, 100, 1, dict(num_disparities=64, block_size=31, uniqueness_ratio=10, prefilter_cap=35)),     # 31^2 x 70 > 65534: sliding-sum kernel
]


@pytest.mark.parametrize("lr", [-1, 1])
@pytest.mark.parametrize("case", range(len(W2931)))
def test_interior_kernel_windows_29_and_31(torch_cuda, pkg, oracle, case, lr):
    from test_gpu_parity import assert_stages_equal, run_engine
    from u96_slam_amd import synth

    w, h, n, kw = W2931[case]
    kw = dict(dict(prefilter_cap=31, texture_threshold=10, speckle_window_size=20, speckle_range=8), **kw)
    kw["disp12_max_diff"] = lr
    L, R = synth.make_batch(90 + case, n, w, h, min(kw["num_disparities"], w // 3))
    eng, ref = run_engine(pkg, oracle, kw, L, R)
    assert_stages_equal(eng, ref, kw)
    assert np.array_equal(eng["disp"], ref["disp"])
    bm = pkg.StereoBM.create(kw["num_disparities"], kw["block_size"])
    bm.setPreFilterCap(kw["prefilter_cap"])
    bm.compute(L[:1], R[:1])
    want = "sad_wide_kernel" if kw["prefilter_cap"] == 35 else "sad_fast_kernel<"
    assert bm.last_kernel().startswith(want), bm.last_kernel()
    if want != "sad_wide_kernel":
        assert f",{kw['block_size']},1," in bm.last_kernel(), bm.last_kernel()
