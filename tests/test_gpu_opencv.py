"""Pin the block-matching output against the REAL cv::StereoBM wherever a box has it (SURVEY.md 8c last row, BASELINE.md
3.1). Probed in round 2: the GPU image has no cv2 and no libopencv_* (profiles/r02_opencv_probe.txt),
so today this skips with that reason; it turns into the parity gate the day an image ships OpenCV. Nothing from
/root/reference or from OpenCV's sources is needed at run time: inputs are the committed golden pair and synthetic frames."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cv2():
    try:
        import cv2

        return cv2
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"OpenCV unavailable on this box (import cv2: {type(e).__name__}): cv::StereoBM parity stays unpinned; the engine is "
                    "checked against the in-repo restatement instead. To settle it off-box: python tools/verify_with_opencv.py "
                    "tests/golden/pin_kit.npz (numpy + cv2 only) -- the kit holds this engine's stage-by-stage outputs for every risk case")


def _cv_bm(cv2, nd, w):
    m = cv2.StereoBM_create(numDisparities=nd, blockSize=w)
    m.setPreFilterCap(31); m.setMinDisparity(0); m.setTextureThreshold(10); m.setUniquenessRatio(10)
    m.setSpeckleWindowSize(50); m.setSpeckleRange(32); m.setDisp12MaxDiff(1)      # main.cpp:204-212
    return m


@pytest.mark.parametrize("w", [9, 21])
def test_reference_pair_against_opencv(pkg, oracle, golden, w):
    cv2 = _cv2()
    print("OpenCV", cv2.__version__)
    ref = _cv_bm(cv2, 64, w).compute(golden["rect_l"], golden["rect_r"])
    bm = pkg.StereoBM.create(64, w)
    bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10)
    bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
    got = bm.compute(golden["rect_l"], golden["rect_r"])
    assert np.array_equal(got, ref), int((got != ref).sum())
    # and the oracle itself gets pinned by the same comparison
    assert np.array_equal(oracle.compute(oracle.make_params(64, w, 31, 0, 10, 10, 50, 32, 1), golden["rect_l"], golden["rect_r"]), ref)


@pytest.mark.parametrize("shape", [(1242, 375, 128, 15, 3), (1920, 1080, 256, 21, 1)])
def test_synthetic_frames_against_opencv(pkg, shape):
    cv2 = _cv2()
    from u96_slam_amd import synth

    W, H, nd, w, n = shape
    L, R = synth.make_batch(0, n, W, H, nd)
    bm = pkg.StereoBM.create(nd, w)
    bm.setPreFilterCap(31); bm.setTextureThreshold(10); bm.setUniquenessRatio(10)
    bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32); bm.setDisp12MaxDiff(1)
    got = bm.compute(L, R)
    m = _cv_bm(cv2, nd, w)
    for i in range(n):
        assert np.array_equal(got[i], m.compute(L[i], R[i])), i


# ---- the pin kit (tools/pin_kit.py): one case per risk item of SURVEY.md A.7 + getValidDisparityROI's "- minDisparity" --------
def _kit():
    import pathlib

    path = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden" / "pin_kit.npz"
    if not path.exists():
        pytest.skip("tests/golden/pin_kit.npz not generated yet (python tools/pin_kit.py on a GPU box)")
    return np.load(path)


def _engine(pkg, q):
    bm = pkg.StereoBM.create(q["num_disparities"], q["block_size"])
    bm.setPreFilterType(q["prefilter_type"]); bm.setPreFilterSize(q["prefilter_size"]); bm.setPreFilterCap(q["prefilter_cap"])
    bm.setMinDisparity(q["min_disparity"]); bm.setTextureThreshold(q["texture_threshold"]); bm.setUniquenessRatio(q["uniqueness_ratio"])
    bm.setSpeckleWindowSize(q["speckle_window_size"]); bm.setSpeckleRange(q["speckle_range"]); bm.setDisp12MaxDiff(q["disp12_max_diff"])
    bm.setROI1((q["roi1_x"], q["roi1_y"], q["roi1_w"], q["roi1_h"])); bm.setROI2((q["roi2_x"], q["roi2_y"], q["roi2_w"], q["roi2_h"]))
    return bm


def _stage(p, st):
    q = dict(p)
    if st == "s0_wta":
        q.update(uniqueness_ratio=0, texture_threshold=0, disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    elif st == "s1_uniq":
        q.update(disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    elif st == "s2_lr":
        q.update(speckle_window_size=0, speckle_range=0)
    return q


def test_engine_reproduces_the_pin_kit(pkg):
    """No OpenCV needed: the committed kit is what THIS engine computes today (a kernel change that moves a single pixel of any
    risk case fails here, before anybody spends an OpenCV box on a stale kit)."""
    kit = _kit()
    fields = [str(f) for f in kit["fields"]]
    for name in kit["names"]:
        name = str(name)
        p = dict(zip(fields, kit[f"{name}/params"].tolist()))
        for st in kit["stages"]:
            got = _engine(pkg, _stage(p, str(st))).compute(kit[f"{name}/left"], kit[f"{name}/right"])
            assert np.array_equal(got, kit[f"{name}/{st}"]), (name, str(st), int((got != kit[f"{name}/{st}"]).sum()))


def test_engine_reproduces_the_alternative_readings_of_the_kit(pkg, monkeypatch):
    """Kit v2: the engine under each bit of SBM_CV_READING (the alternative reading of a cv::StereoBM behaviour nobody could pin)
    still produces what the kit stores for the cases that tell the readings apart."""
    kit = _kit()
    fields = [str(f) for f in kit["fields"]]
    n = 0
    for b, cs in zip(kit["risk_bits"].tolist(), kit["risk_cases"].tolist()):
        monkeypatch.setenv("SBM_CV_READING", str(int(b)))
        for name in str(cs).split(","):
            p = dict(zip(fields, kit[f"{name}/params"].tolist()))
            for st in kit["stages"]:
                got = _engine(pkg, _stage(p, str(st))).compute(kit[f"{name}/left"], kit[f"{name}/right"])
                want = kit[f"{name}/alt{b}/{st}"]
                assert np.array_equal(got, want), (name, int(b), str(st), int((got != want).sum()))
                n += 1
    assert n >= 4 * 9


def test_pin_kit_against_opencv():
    """The one-command pin, run in-tree where a box has cv2: every risk case, first differing stage reported."""
    cv2 = _cv2()
    import importlib.util
    import pathlib

    root = pathlib.Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("verify_with_opencv", root / "tools" / "verify_with_opencv.py")
    ver = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ver)
    kit = _kit()
    fields = [str(f) for f in kit["fields"]]
    bad = []
    for name in kit["names"]:
        name = str(name)
        p = dict(zip(fields, kit[f"{name}/params"].tolist()))
        for st in kit["stages"]:
            got = ver.make_bm(cv2, ver.stage_params(p, str(st))).compute(kit[f"{name}/left"], kit[f"{name}/right"])
            if not np.array_equal(got, kit[f"{name}/{st}"]):
                bad.append((name, str(st), int((got != kit[f"{name}/{st}"]).sum())))
                break
    assert not bad, bad
