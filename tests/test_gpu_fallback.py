"""The two-accumulator fallback build of the interior SAD kernel (sbm_sad_fast_pp.hip: what runs when the device self-test of
the in-place v_mqsad accumulate fails, selected here with SBM_FAST_INPLACE=0). Round 5 reduced it to the 64-disparity layouts
with masked-count kernels only and it is the last user of the register-staged strip, so it gets its own parity sweep: every
layout (one to four cooperating wavefronts), exact and masked counts, 3- and 1-column sums, one-pair and batched launches.
The switch is read once per process, hence the subprocess."""
import json
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]

SCRIPT = r"""
import json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/oracle")
import numpy as np, torch
import _pkg, sbm_oracle
pkg = _pkg.load()
from u96_slam_amd import synth
out = []
for W, H, nd, w, n in [(320, 96, 64, 21, 5), (400, 80, 128, 15, 9), (640, 120, 256, 21, 2), (333, 77, 48, 11, 7), (300, 70, 96, 27, 17),
                       (500, 90, 192, 19, 3), (420, 80, 112, 15, 40), (360, 70, 16, 5, 1), (400, 90, 160, 25, 2), (640, 480, 64, 21, 1),
                       (700, 60, 320, 15, 2), (400, 90, 64, 29, 2), (900, 66, 384, 9, 1)]:   # (the last three: beyond 256 disparities / 27 x 27 the fallback build hands over to the sliding-sum kernel)
    L, R = synth.make_batch(3, n, W, H, nd)
    bm = pkg.StereoBM.create(nd, w)
    bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(30); bm.setSpeckleRange(16)
    p = sbm_oracle.make_params(nd, w, 31, 0, 10, 10, 30, 16, 1)
    got = bm.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    ref = sbm_oracle.compute_batch(p, L, R)
    out.append({"case": [W, H, nd, w, n], "ok": bool(np.array_equal(got, ref)), "kernel": bm.last_kernel()})
print(json.dumps(out))
"""


@pytest.mark.gpu
def test_fallback_build_parity_sweep():
    import os

    env = dict(os.environ, SBM_FAST_INPLACE="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": str(ROOT)}], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
    assert len(res) == 13
    for e in res[-3:]:     # the documented cliff (include/sbm.h): right results, from the sliding-sum kernel, and the name says why
        assert e["kernel"] == "sad_wide_kernel [in-place accumulate unavailable]" and e["ok"], e
    res = res[:-3]
    for e in res:
        assert e["kernel"].startswith("sad_fast_pp_kernel<64,"), e
        assert e["ok"], e
    assert {e["kernel"].split(",")[1] for e in res} == {"1", "2", "3", "4"}      # every layout of the fallback ran


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1242, 375, 128, 15), (640, 480, 64, 21), (1920, 1080, 256, 21), (1242, 375, 128, 19), (800, 600, 192, 9)])
def test_partial_chip_launches_bit_exact(shape):
    """Launches that do not fill the chip take many short row segments (down to 8 rows, up to 64, tapered) and, for one or two
    pairs, the split layouts: 1, 2, 3, 5 and 8 full-size pairs against the oracle, whole maps."""
    import numpy as np
    import torch

    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
    import _pkg
    import sbm_oracle

    pkg = _pkg.load()
    from u96_slam_amd import synth

    W, H, nd, w = shape
    p = sbm_oracle.make_params(nd, w, 31, 0, 10, 10, 50, 32, 1)
    Lall, Rall = synth.make_batch(40, 8, W, H, nd)
    ref = sbm_oracle.compute_batch(p, Lall, Rall)
    kernels = set()
    for n in (1, 2, 3, 5, 8):
        if W * H * nd * n > 3.0e9:        # (keeps 1080p at 256 disparities to 1..5 pairs)
            continue
        bm = pkg.StereoBM.create(nd, w)
        bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32)
        got = bm.compute_device(torch.from_numpy(Lall[:n]).cuda(), torch.from_numpy(Rall[:n]).cuda()).cpu().numpy()
        assert np.array_equal(got, ref[:n]), (shape, n, int((got != ref[:n]).sum()), bm.last_kernel())
        kernels.add(bm.last_kernel())
    assert len(kernels) >= 2 or nd > 128, kernels      # the split layout for the smallest launches, the regular one beyond


@pytest.mark.gpu
def test_one_small_pair_at_256_disparities():
    """One 640x480 pair at 256 disparities cannot fill the chip with two-wavefront workgroups: the launch splits the disparities
    over four 64-disparity wavefronts (masked-count kernel: <64,4> has no exact one). Whole map against the oracle."""
    import numpy as np
    import torch

    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
    import _pkg
    import sbm_oracle

    pkg = _pkg.load()
    from u96_slam_amd import synth

    L, R = synth.make_batch(7, 1, 640, 480, 256)
    bm = pkg.StereoBM.create(256, 21)
    bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(50); bm.setSpeckleRange(32)
    got = bm.compute_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    assert bm.last_kernel().startswith("sad_fast_kernel<64,4,7,3,false,true>"), bm.last_kernel()
    p = sbm_oracle.make_params(256, 21, 31, 0, 10, 10, 50, 32, 1)
    assert np.array_equal(got, sbm_oracle.compute_batch(p, L, R))
