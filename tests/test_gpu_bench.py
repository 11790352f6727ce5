"""bench.py contract on the GPU box: the N=1 line carries roofline + cpu_baseline (with the OpenCV probe outcome), and
`python bench.py --gpus 2` launches its own ranks (gloo here: both ranks share the one GPU of the test box, so the
numbers are meaningless -- the control flow, the chunked scatter/gather and the JSON are what is checked)."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def _run(args, env=None, timeout=600):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_cpu_baseline():
    j = _run(["--steps", "5", "--warmup", "2", "--pairs", "8", "--cpu-sample", "4", "--check"])
    assert j["n_gpus"] == 1 and j["unit"] == "Mpix-disparities/s" and j["dtype"] == "u8" and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["limiter"] == "valu" and 0 < j["roofline"]["frac"] < 1
    assert j["roofline_prefilter"]["device_copy_GBps"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["bit_exact_vs_gpu"] is True
    assert cb["opencv"].startswith(("cv2 ", "unavailable"))      # the probe outcome is always recorded


@pytest.mark.parametrize("extra", [[], ["--scatter", "--chunk", "3"]])
def test_two_ranks_self_launched(extra):
    j = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "8"] + extra, env={"SBM_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["config"]["global_pairs_per_step"] == 16 and j["value"] > 0
    assert ("chunked" in j["config"]["parallelism"]) == bool(extra)
