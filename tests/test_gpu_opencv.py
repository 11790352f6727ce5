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
        pytest.skip(f"OpenCV unavailable on this box (import cv2: {type(e).__name__}): cv::StereoBM parity stays unpinned; "
                    "the engine is checked against the in-repo restatement instead")


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
