"""The pin kit (tools/pin_kit.py -> tests/golden/pin_kit.npz): inputs, parameter blocks and the ENGINE's stage-by-stage outputs
for one case per bit-exactness risk of SURVEY.md A.7. CPU-side: the kit's hashes are what the manifest says, the oracle
reproduces every stored output (kit == engine == oracle), and the stand-alone verifier's helpers agree with the generator's.
Settling kit == cv::StereoBM needs OpenCV: `python tools/verify_with_opencv.py tests/golden/pin_kit.npz` on any box that has it."""
import hashlib
import importlib.util
import pathlib

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
KIT = ROOT / "tests" / "golden" / "pin_kit.npz"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def kit():
    if not KIT.exists():
        pytest.skip("tests/golden/pin_kit.npz not generated yet (python tools/pin_kit.py on a GPU box)")
    return np.load(KIT)


def test_manifest_hashes(kit):
    lines = KIT.with_suffix(".sha256").read_text().splitlines()
    want = {l.split()[1]: l.split()[0] for l in lines}
    assert sorted(want) == sorted(str(n) for n in kit["names"])
    for name in want:
        h = hashlib.sha256()
        keys = ["left", "right", "params", *[str(s) for s in kit["stages"]]]
        for b, cs in zip(kit["risk_bits"].tolist(), kit["risk_cases"].tolist()):      # kit v2: the alternative readings, in bit order
            if name in str(cs).split(","):
                keys += [f"alt{b}/{s}" for s in kit["stages"]]
        for key in keys:
            h.update(np.ascontiguousarray(kit[f"{name}/{key}"]).tobytes())
        assert h.hexdigest() == want[name], name


def test_oracle_reproduces_every_stage(kit, oracle):
    gen = _load(ROOT / "tools" / "pin_kit.py", "pin_kit_gen")
    ver = _load(ROOT / "tools" / "verify_with_opencv.py", "pin_kit_ver")
    fields = [str(f) for f in kit["fields"]]
    assert tuple(fields) == gen.FIELDS
    n = 0
    for name in kit["names"]:
        name = str(name)
        p = dict(zip(fields, kit[f"{name}/params"].tolist()))
        L, R = kit[f"{name}/left"], kit[f"{name}/right"]
        for st in kit["stages"]:
            q = ver.stage_params(p, str(st))          # the verifier's own staging, not the generator's
            po = oracle.make_params(q["num_disparities"], q["block_size"], q["prefilter_cap"], q["min_disparity"], q["texture_threshold"],
                                    q["uniqueness_ratio"], q["speckle_window_size"], q["speckle_range"], q["disp12_max_diff"],
                                    q["prefilter_type"], q["prefilter_size"], (q["roi1_x"], q["roi1_y"], q["roi1_w"], q["roi1_h"]),
                                    (q["roi2_x"], q["roi2_y"], q["roi2_w"], q["roi2_h"]))
            assert np.array_equal(oracle.compute(po, L, R), kit[f"{name}/{st}"]), (name, str(st))
            n += 1
            for b, cs in zip(kit["risk_bits"].tolist(), kit["risk_cases"].tolist()):   # kit v2: the oracle under the alternative reading
                if name in str(cs).split(","):
                    with oracle.reading(int(b)):
                        assert np.array_equal(oracle.compute(po, L, R), kit[f"{name}/alt{b}/{st}"]), (name, str(st), int(b))
    assert n >= 4 * 20


def test_every_risk_has_a_case_that_tells_the_readings_apart(kit):
    """Kit v2: for each bit of SBM_CV_READING at least one case's stored outputs differ between the default and the alternative
    reading -- otherwise running the verifier could not name the reading."""
    assert kit["risk_bits"].tolist() == [1, 2, 4, 8, 16]
    for b, cs in zip(kit["risk_bits"].tolist(), kit["risk_cases"].tolist()):
        apart = 0
        for name in str(cs).split(","):
            apart += any(not np.array_equal(kit[f"{name}/alt{b}/{st}"], kit[f"{name}/{st}"]) for st in kit["stages"])
        assert apart >= 1, b


def test_kit_covers_the_risk_list(kit):
    names = {str(n) for n in kit["names"]}
    for must in ("ref_pair_w21_callsite", "mind_neg8_rois", "mind_pos4_rois", "cost_w17_cap63", "cost_w23_cap31", "odd_height",
                 "speckle_range1", "speckle_range16", "lr_ties_d12_1", "prefilter_norm_9"):
        assert must in names
    # the stages really separate the filters somewhere
    assert not np.array_equal(kit["speckle_range1/s3_full"], kit["speckle_range16/s3_full"])
    assert not np.array_equal(kit["ref_pair_w21_callsite/s1_uniq"], kit["ref_pair_w21_callsite/s0_wta"])
    assert not np.array_equal(kit["ref_pair_w21_callsite/s2_lr"], kit["ref_pair_w21_callsite/s1_uniq"])
    assert not np.array_equal(kit["ref_pair_w21_callsite/s3_full"], kit["ref_pair_w21_callsite/s2_lr"])


def test_verifier_runs_end_to_end_with_a_stand_in(kit, oracle, monkeypatch, capsys):
    """tools/verify_with_opencv.py against a stand-in `cv2` whose StereoBM is the oracle: the script's own logic (parameter
    transport through the cv2 setter names, staging, first-difference report, exit code) runs here; it says nothing about the
    real OpenCV, and a deliberately broken stand-in must be caught at the right stage."""
    import sys
    import types

    class FakeBM:
        def __init__(self, numDisparities, blockSize, break_lr=False):
            self.p = dict(num_disparities=numDisparities, block_size=blockSize, prefilter_type=1, prefilter_size=9, prefilter_cap=31,
                          min_disparity=0, texture_threshold=10, uniqueness_ratio=15, speckle_window_size=0, speckle_range=0,
                          disp12_max_diff=-1, roi1=(0, 0, 0, 0), roi2=(0, 0, 0, 0))
            self.break_lr = break_lr

        def setPreFilterType(self, v): self.p["prefilter_type"] = v
        def setPreFilterSize(self, v): self.p["prefilter_size"] = v
        def setPreFilterCap(self, v): self.p["prefilter_cap"] = v
        def setMinDisparity(self, v): self.p["min_disparity"] = v
        def setTextureThreshold(self, v): self.p["texture_threshold"] = v
        def setUniquenessRatio(self, v): self.p["uniqueness_ratio"] = v
        def setSpeckleWindowSize(self, v): self.p["speckle_window_size"] = v
        def setSpeckleRange(self, v): self.p["speckle_range"] = v
        def setDisp12MaxDiff(self, v): self.p["disp12_max_diff"] = v if not self.break_lr else -1
        def setROI1(self, r): self.p["roi1"] = tuple(r)
        def setROI2(self, r): self.p["roi2"] = tuple(r)

        def compute(self, L, R):
            q = self.p
            po = oracle.make_params(q["num_disparities"], q["block_size"], q["prefilter_cap"], q["min_disparity"], q["texture_threshold"],
                                    q["uniqueness_ratio"], q["speckle_window_size"], q["speckle_range"], q["disp12_max_diff"],
                                    q["prefilter_type"], q["prefilter_size"], q["roi1"], q["roi2"])
            return oracle.compute(po, L, R)

    ver = _load(ROOT / "tools" / "verify_with_opencv.py", "pin_kit_ver2")
    for broken in (False, True):
        fake = types.ModuleType("cv2")
        fake.__version__ = "stand-in"
        fake.StereoBM_create = lambda numDisparities=0, blockSize=21, _b=broken: FakeBM(numDisparities, blockSize, _b)
        monkeypatch.setitem(sys.modules, "cv2", fake)
        monkeypatch.setattr(sys, "argv", ["verify_with_opencv.py", str(KIT)])
        rc = ver.main()
        out = capsys.readouterr().out
        if not broken:
            assert rc == 0 and "ALL CASES AGREE" in out
        else:
            assert rc == 1 and "FIRST DIFFERENCE at stage s2_lr" in out and "stage s0_wta" not in out and "stage s1_uniq" not in out
    # a stand-in that implements ALTERNATIVE readings (speckleRange x 16 and the later-x LR tie): the verifier must name the bits
    fake = types.ModuleType("cv2")
    fake.__version__ = "stand-in (alternative readings 4 + 16)"
    fake.StereoBM_create = lambda numDisparities=0, blockSize=21: FakeBM(numDisparities, blockSize, False)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    monkeypatch.setattr(sys, "argv", ["verify_with_opencv.py", str(KIT)])
    with oracle.reading(4 | 16):
        rc = ver.main()
    out = capsys.readouterr().out
    assert rc == 1 and "ALTERNATIVE READING bit 4 matches" in out and "ALTERNATIVE READING bit 16 matches" in out
    assert "default to 20" in out     # (other cases differ too -- a consequence of the same readings, which the verifier says)
