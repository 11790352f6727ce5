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
    # the launched SAD kernel is named (template tuple) and the per-step distribution is reported
    assert j["roofline"]["kernel"].startswith("sad_fast_kernel<128,1,5,3,true,true>")
    assert 0 < j["ms_per_step_min"] <= j["ms_per_step_median"]
    # counters from the committed profile are only attached to the kernel they were measured on
    assert (j["roofline"]["traffic"] is not None) != ("traffic_reason" in j["roofline"]) or j["roofline"]["traffic"] is None


def test_host_feed_line():
    j = _run(["--steps", "4", "--warmup", "1", "--pairs", "32", "--no-cpu-baseline", "--feed", "host"])
    hf = j["host_feed"]
    assert "pinned host memory" in j["config"]["parallelism"] and hf["h2d_GBps"] > 0 and hf["d2h_GBps"] > 0
    assert 0.0 <= hf["overlap_frac"] <= 1.0 and hf["resident_compute_ms"] > 0


@pytest.mark.parametrize("extra", [[], ["--scatter", "--chunk", "3"]])
def test_two_ranks_self_launched(extra):
    j = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "8"] + extra, env={"SBM_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["config"]["global_pairs_per_step"] == 16 and j["value"] > 0
    assert ("chunked (" in j["config"]["parallelism"]) == bool(extra)
    # the default line carries the multi-GPU evidence without any flag: rank identity + the chunked scatter/gather
    assert j["rccl"]["ranks_seen"] == 2 and j["rccl"]["distinct_devices"] >= 1
    if not extra:
        assert j["multi_cxx"]["equals_resident_shard"] is True and j["multi_cxx"]["pairs_per_call"] == 16, j["multi_cxx"]
        sg = j["scatter_gather"]
        assert sg["ms_per_step"] > 0 and sg["value"] > 0 and sg["chunk"] == 8 and sg["backend"] == "gloo" and 0 <= sg["overlap_frac"] <= 1
        assert j["ms_per_step_median"] >= j["ms_per_step_min"] > 0


def test_two_ranks_under_torchrun_like_the_driver():
    """The driver launches N > 1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: the ranks are
    torchrun workers (TORCHELASTIC_* in their environment), and the scatter/gather children they start must form their own
    process group all the same."""
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "8",
                        "--no-cpu-baseline", "--sg-timeout", "120"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SBM_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["rccl"]["ranks_seen"] == 2
    assert "error" not in j["scatter_gather"] and j["scatter_gather"]["ms_per_step"] > 0, j["scatter_gather"]


def test_eight_ranks_under_torchrun_on_one_gpu():
    """The driver's N = 8 launch line on a one-GPU box (gloo: the eight ranks share the device, the numbers mean nothing): rank
    identity over all eight ranks, the chunked scatter/gather leg in its child processes, the one-process C++ leg
    (sbm_compute_batch_multi over the visible devices, checked against rank 0's resident shard) and a wall time far inside any
    driver limit -- so that the first real 8-GPU run is not also the first run of this control flow."""
    import socket
    import time

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--pairs", "8",
                        "--no-cpu-baseline", "--sg-timeout", "150"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, SBM_BENCH_BACKEND="gloo"))
    wall = time.perf_counter() - t0
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["global_pairs_per_step"] == 64 and j["value"] > 0
    assert j["rccl"]["ranks_seen"] == 8 and j["rccl"]["backend"] == "gloo"
    sg = j["scatter_gather"]
    assert "error" not in sg and sg["ms_per_step"] > 0 and sg["value"] > 0, sg
    mc = j["multi_cxx"]
    assert "error" not in mc, mc
    assert mc["api"] == "sbm_compute_batch_multi" and mc["devices"] >= 1 and mc["pairs_per_call"] == 64 and mc["equals_resident_shard"] is True
    assert wall < 420, wall


@pytest.mark.parametrize("fault", ["hang", "crash"])
def test_scatter_gather_fault_cannot_lose_the_value(fault):
    """The point-to-point leg runs in child processes of the timed ranks with a wall-clock limit: a rank that hangs (or dies)
    in there costs the timeout and an error field -- rc 0, one JSON line and the shards-resident value survive."""
    j = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "8", "--no-cpu-baseline", "--sg-timeout", "40"],
             env={"SBM_BENCH_BACKEND": "gloo", "SBM_BENCH_SG_FAULT": fault})
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["rccl"]["ranks_seen"] == 2
    assert "error" in j["scatter_gather"] and ("timeout" in j["scatter_gather"]["error"] or "exited" in j["scatter_gather"]["error"])


def test_first_steps_and_prewarm_are_disclosed():
    j = _run(["--steps", "12", "--warmup", "2", "--pairs", "8", "--no-cpu-baseline"])
    assert j["prewarm_s"] == 0.3 and j["prewarm_steps"] >= 8 and j["warmup"] == 2 and j["steps"] == 12
    assert j["ms_per_step_first5"] > 0 and j["ms_per_step_after5"] > 0 and j["engine_library"] == "libsbm_hip.so"


def test_host_submissions_do_not_pile_up():
    """A feeder that never calls synchronize(): wait_host() hands every delivered batch back."""
    import numpy as np

    sys.path.insert(0, str(ROOT))
    import _pkg

    pkg = _pkg.load()
    from u96_slam_amd import synth

    W, H, nd, B = 160, 48, 32, 2
    bm = pkg.StereoBM.create(nd, 9, device=0)
    L, R = synth.make_batch(0, B, W, H, nd)
    ref = bm.compute(L, R)
    outs = []
    for k in range(100):
        out = np.empty((B, H, W), np.int16)
        bm.submit_host(L, R, out)
        outs.append(out)
        if k >= 2:
            bm.wait_host()
            assert np.array_equal(outs[k - 2], ref)
        assert len(bm._host_inflight) <= 3
    # a synchronous host call between submissions and their waits must not disturb the queue:
    # it re-sizes its own staging only
    big = synth.make_batch(5, 3, W + 16, H + 8, nd)
    bm.compute(big[0], big[1])
    bm.wait_host(); bm.wait_host()
    assert np.array_equal(outs[98], ref) and np.array_equal(outs[99], ref) and len(bm._host_inflight) == 0


def test_async_dense_feed_matches_the_synchronous_call():
    """sbm_submit_dense / sbm_wait_oldest: three batches in flight on two device staging sets, results identical to
    sbm_compute_batch and to the oracle, buffers of different submissions never mix."""
    import numpy as np
    import torch

    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
    import _pkg
    import sbm_oracle

    pkg = _pkg.load()
    from u96_slam_amd import synth

    W, H, nd, B = 200, 64, 32, 5
    bm = pkg.StereoBM.create(nd, 9, device=0)
    bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(30); bm.setSpeckleRange(16)
    p = sbm_oracle.make_params(nd, 9, 31, 0, 10, 10, 30, 16, 1)
    batches = []
    for k in range(7):     # more submissions than queue slots
        L, R = synth.make_batch(100 * k, B, W, H, nd)
        hl, hr = torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()
        hd = torch.full((B, H, W), 12345, dtype=torch.int16).pin_memory()
        batches.append((L, R, hl, hr, hd))
    for k, (L, R, hl, hr, hd) in enumerate(batches):
        bm.submit_host(hl.numpy(), hr.numpy(), hd.numpy())
        if k >= 2:
            bm.wait_host()
            done = batches[k - 2]
            assert np.array_equal(done[4].numpy(), sbm_oracle.compute_batch(p, done[0], done[1]))
    bm.synchronize()         # drains the last two
    for L, R, hl, hr, hd in batches:
        ref = sbm_oracle.compute_batch(p, L, R)
        assert np.array_equal(hd.numpy(), ref)
        assert np.array_equal(bm.compute(L, R), ref)
    bm.wait_host()           # nothing outstanding: returns at once


@pytest.mark.parametrize("env,args,kernel", [
    ({"SBM_FAST_INPLACE": "0"}, ["--pairs", "8"], "sad_fast_pp_kernel<64,2,5,3,false,true> pfshift=2"),   # two-accumulator fallback build (64-disparity layouts, masked-count kernels only)
    ({}, ["--pairs", "1"], "sad_fast_kernel<64,2,5,3,true,true> pfshift=2"),                              # one pair per call: the disparities split over two cooperating wavefronts
    ({"SBM_FAST_PFSHIFT": "0"}, ["--pairs", "8"], "sad_fast_kernel<128,1,5,3,true,true> pfshift=0"),      # unscaled planes, plain key search
    ({"SBM_FAST_PFSHIFT": "1"}, ["--pairs", "8"], "sad_fast_kernel<128,1,5,3,true,true> pfshift=0"),      # w 15 kernels hold the two-bit variant only
    ({"SBM_FAST_CS3": "0"}, ["--pairs", "8"], "sad_fast_kernel<128,1,5,3,true,true> pfshift=2"),          # plain strips only (no column stride 3)
])
def test_engine_variants_are_bit_exact(env, args, kernel):
    """Every selectable variant of the interior kernel against the oracle on the bench workload (the engine reads these
    switches once per process, hence the subprocess): the fallback that runs when the device self-test of the in-place
    v_mqsad accumulate fails must be as exact as the default."""
    j = _run(["--steps", "2", "--warmup", "1", "--cpu-sample", "8", "--check"] + args, env=env)
    assert j["roofline"]["kernel"] == kernel
    assert j["cpu_baseline"]["bit_exact_vs_gpu"] is True


def test_256_disparities_two_cooperating_wavefronts_bit_exact():
    """BASELINE configs[2]'s layout: two 128-disparity wavefronts, LDS-direct staging (the four 64-disparity wavefronts that
    small launches take instead: tests/test_gpu_fallback.py::test_one_small_pair_at_256_disparities)."""
    j = _run(["--steps", "2", "--warmup", "1", "--pairs", "4", "--cpu-sample", "1", "--check", "--workload", "fhd", "--prewarm-s", "0"])
    assert j["roofline"]["kernel"] == "sad_fast_kernel<128,2,7,3,true,true> pfshift=1" and j["cpu_baseline"]["bit_exact_vs_gpu"] is True


def test_reference_window_uses_one_tag_bit():
    j = _run(["--steps", "2", "--warmup", "1", "--pairs", "8", "--cpu-sample", "8", "--check", "--workload", "ref640"])
    assert j["roofline"]["kernel"] == "sad_fast_kernel<64,1,7,3,true,true> pfshift=1" and j["cpu_baseline"]["bit_exact_vs_gpu"] is True
