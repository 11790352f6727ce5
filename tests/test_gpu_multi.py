"""The C++ caller's multi-engine path (sbm_compute_batch_multi) and the thread-safety promise of include/sbm.h, on whatever
the box has: one handle per visible device, and -- so that a one-GPU box still exercises the block partition and the shared
state (parked-handle pool, device self-test, kernel-name buffer) -- two engines / two host threads on device 0."""
import pathlib
import subprocess
import threading

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


def _build(tmp_path, pkg):
    exe = tmp_path / "multi_main"
    lib = pkg.library_path()
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
                        str(ROOT / "tests" / "cpp" / "multi_main.cpp"), "-o", str(exe), str(lib), "-L/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    return exe, r


@pytest.mark.gpu
@pytest.mark.parametrize("per_dev", [1, 2])
def test_cpp_multi_engine_and_two_threads(tmp_path, pkg, oracle, golden, per_dev):
    """tests/cpp/multi_main.cpp: whole batch on one handle == sbm_compute_batch_multi over one (two) handle(s) per visible
    device == two host threads with their own handles calling sbm_compute pair by pair; and all of it == the oracle."""
    exe, r = _build(tmp_path, pkg)
    assert r.returncode == 0, r.stderr
    L0, R0 = golden["rect_l"], golden["rect_r"]
    n = 7                                          # odd: uneven blocks and halves
    frames_l = np.stack([np.roll(L0, 5 * i, axis=1) if i % 2 == 0 else np.roll(L0[::-1], 3 * i, axis=1) for i in range(n)])
    frames_r = np.stack([np.roll(R0, 5 * i, axis=1) if i % 2 == 0 else np.roll(R0[::-1], 3 * i, axis=1) for i in range(n)])
    (tmp_path / "l.raw").write_bytes(frames_l.tobytes())
    (tmp_path / "r.raw").write_bytes(frames_r.tobytes())
    run = subprocess.run([str(exe), "640", "480", str(n), str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(tmp_path), str(per_dev)],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "OK" in run.stdout
    p = oracle.make_params(64, 21, 31, 0, 10, 10, 50, 32, 1)      # the parameters of main.cpp:204-212
    for name in ("single", "multi", "threads"):
        got = np.frombuffer((tmp_path / f"{name}.raw").read_bytes(), np.int16).reshape(n, 480, 640)
        for i in (0, 3, n - 1):
            ref = oracle.compute(p, frames_l[i], frames_r[i])
            assert np.array_equal(got[i], ref), (name, i, int((got[i] != ref).sum()))


@pytest.mark.gpu
def test_compute_multi_blocks_python(pkg, oracle, torch_cuda):
    """compute_multi over engines with DIFFERENT parameter blocks: every block equals what its own engine computes alone
    (handles keep their parameters; blocks are contiguous, [n k / K, n (k + 1) / K))."""
    from u96_slam_amd import synth

    torch = torch_cuda
    n, W, H = 5, 320, 96
    L, R = synth.make_batch(11, n, W, H, 64)
    ndev = torch.cuda.device_count()
    engines = []
    for k in range(max(3, ndev)):
        bm = pkg.StereoBM.create(64 if k % 2 == 0 else 32, 15 if k % 3 else 9, device=k % ndev)
        bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(30); bm.setSpeckleRange(16)
        engines.append(bm)
    Lp, Rp = torch.from_numpy(L).pin_memory().numpy(), torch.from_numpy(R).pin_memory().numpy()
    out = torch.empty((n, H, W), dtype=torch.int16).pin_memory().numpy()
    pkg.compute_multi(engines, Lp, Rp, out)
    K = len(engines)
    for k, bm in enumerate(engines):
        b0, b1 = n * k // K, n * (k + 1) // K
        if b1 > b0:
            p = oracle.make_params(bm.getNumDisparities(), bm.getBlockSize(), 31, 0, 10, 10, 30, 16, 1)
            for i in range(b0, b1):
                assert np.array_equal(out[i], oracle.compute(p, L[i], R[i])), (k, i)
    with pytest.raises(pkg.StereoBMError):
        pkg.compute_multi([engines[0], engines[0]], Lp, Rp, out)
    # fewer pairs than engines: the first engines get empty blocks, the last one the pair
    one = torch.empty((1, H, W), dtype=torch.int16).pin_memory().numpy()
    pkg.compute_multi(engines, Lp[:1], Rp[:1], one)
    bm = engines[-1]
    p = oracle.make_params(bm.getNumDisparities(), bm.getBlockSize(), 31, 0, 10, 10, 30, 16, 1)
    assert np.array_equal(one[0], oracle.compute(p, L[0], R[0]))


@pytest.mark.gpu
def test_every_visible_device_gets_a_handle(pkg, oracle, torch_cuda):
    """A handle on every visible device computes the same map (device != 0 only exists on multi-GPU boxes; the loop still
    runs on one GPU)."""
    from u96_slam_amd import synth

    torch = torch_cuda
    L, R = synth.make_batch(5, 1, 400, 80, 64)
    p = oracle.make_params(64, 15, 31, 0, 10, 15, 0, 0, -1)
    ref = oracle.compute(p, L[0], R[0])
    for d in range(torch.cuda.device_count()):
        bm = pkg.StereoBM.create(64, 15, device=d)
        assert np.array_equal(bm.compute(L[0], R[0]), ref), d


@pytest.mark.gpu
def test_two_python_threads_two_engines(pkg, oracle, torch_cuda):
    """Two host threads, each creating / using / dropping its own engines on device 0 (ctypes releases the GIL inside the
    C calls, so the library's shared state -- handle pool, once-per-device self-test -- really is entered concurrently)."""
    from u96_slam_amd import synth

    L, R = synth.make_batch(9, 4, 320, 96, 64)
    want = {}
    for w in (9, 15):
        p = oracle.make_params(64, w, 31, 0, 10, 10, 30, 16, 1)
        want[w] = [oracle.compute(p, L[i], R[i]) for i in range(4)]
    errs = []

    def worker(w):
        try:
            for rep in range(6):
                bm = pkg.StereoBM.create(64, w)
                bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(30); bm.setSpeckleRange(16)
                for i in range(4):
                    if not np.array_equal(bm.compute(L[i], R[i]), want[w][i]):
                        errs.append((w, rep, i))
                del bm
        except Exception as e:      # noqa: BLE001
            errs.append((w, repr(e)))

    ts = [threading.Thread(target=worker, args=(w,)) for w in (9, 15)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("zc", ["1", "0"])
def test_small_host_call_paths_agree(pkg, oracle, torch_cuda, monkeypatch, zc):
    """sbm_compute on pageable and on pinned caller memory, through the copy kernel + flag (SBM_HOST_ZEROCOPY=1, pageable only)
    and through the D2H copy: same maps; odd sizes exercise the 2-byte tail of the copy kernel; repeated calls its sequence flag."""
    from u96_slam_amd import synth

    torch = torch_cuda
    monkeypatch.setenv("SBM_HOST_ZEROCOPY", zc)
    for W, H, nd in ((321, 97, 32), (640, 480, 64), (200, 64, 16)):
        L, R = synth.make_batch(21, 2, W, H, nd)
        bm = pkg.StereoBM.create(nd, 9)
        bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1); bm.setSpeckleWindowSize(30); bm.setSpeckleRange(16)
        p = oracle.make_params(nd, 9, 31, 0, 10, 10, 30, 16, 1)
        for rep in range(3):
            for i in range(2):
                ref = oracle.compute(p, L[i], R[i])
                out = np.full((H, W), 12345, np.int16)
                assert np.array_equal(bm.compute(L[i], R[i], out), ref), (W, H, rep, i, "pageable")
                pin = torch.full((H, W), 12345, dtype=torch.int16).pin_memory().numpy()
                assert np.array_equal(bm.compute(L[i], R[i], pin), ref), (W, H, rep, i, "pinned")
        both = bm.compute(L, R)                       # a two-pair batch through the same entry point
        for i in range(2):
            assert np.array_equal(both[i], oracle.compute(p, L[i], R[i]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(400, 90, 128, 15, 3, 0), (640, 120, 64, 21, 1, 0), (420, 80, 112, 15, 5, 0), (500, 90, 192, 21, 2, 0),
                                   (400, 90, 128, 19, 2, 0), (640, 120, 256, 21, 2, 0), (360, 80, 48, 11, 1, -8), (400, 90, 64, 27, 3, 4),
                                   (800, 60, 400, 9, 2, 0), (900, 66, 512, 27, 1, 0)])
def test_filtered_pixels_store_cost_ffff(pkg, torch_cuda, shape):
    """The range-test-free path of lrcheck16_kernel relies on an invariant that spans files: every producer of the
    16-bit cost plane -- LDS-direct strips of every layout, border wavefronts -- stores cost 0xffff with every FILTERED pixel
    of the computed region [lofs, lofs + xend) x [row0, row1) and real costs <= 65534. Checked directly here, not through a
    downstream mismatch: single-wavefront, cooperating, masked-count and one-pair (split) layouts, 1- and 3-column sums, and --
    beyond 256 disparities -- the border columns that sbm_sad_wide.hip writes into the same 16-bit plane."""
    from u96_slam_amd import synth

    W, H, nd, w, n, mind = shape
    L, R = synth.make_batch(31, n, W, H, nd)
    bm = pkg.StereoBM.create(nd, w)
    bm.setMinDisparity(mind); bm.setUniquenessRatio(10); bm.setTextureThreshold(10); bm.setDisp12MaxDiff(1)
    bm.compute(L, R)
    pre = bm.debug_fetch(3, n, H, W)
    cost = bm.debug_fetch(2, n, H, W)
    filtered = (mind - 1) * 16
    lofs, rofs = max(nd - 1 + mind, 0), -min(nd - 1 + mind, 0)
    xend = min(W - rofs - nd + 1, W - lofs)
    w2 = w // 2
    region = (slice(None), slice(w2, H - w2), slice(lofs, lofs + xend))
    p, c = pre[region], cost[region]
    assert bm.last_kernel().startswith("sad_fast_kernel"), bm.last_kernel()
    assert ((p == filtered) == (c == 0xFFFF)).all(), (int(((p == filtered) & (c != 0xFFFF)).sum()), int(((p != filtered) & (c == 0xFFFF)).sum()))
    assert (c[p != filtered] <= 65534).all() and (p != filtered).mean() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("outstanding", [1, 2, 3])
def test_destroy_with_outstanding_submissions(pkg, oracle, torch_cuda, outstanding):
    """The contract of sbm_destroy() and the asynchronous feed (include/sbm.h): copies that are already queued
    finish, the NEWEST submission's maps are dropped, and destroy never starts a write into caller memory -- checked through
    the bare C-ABI with 1, 2 and 3 submissions nobody waited for. The Python mirror's close() (and the C++ adaptor's
    destructor) drain first, so nothing is dropped there."""
    import ctypes
    import torch
    from u96_slam_amd import synth, stereobm

    W, H, nd, B = 200, 64, 32, 2
    Lb = stereobm.load_library()
    prm = stereobm.SbmParams()
    Lb.sbm_params_default(ctypes.byref(prm), nd, 9)
    prm.uniqueness_ratio, prm.disp12_max_diff = 10, 1
    op = oracle.make_params(nd, 9, 31, 0, 10, 10, 0, 0, 1)
    h = ctypes.c_void_p()
    assert Lb.sbm_create(ctypes.byref(h), ctypes.byref(prm), 0) == 0
    sentinel = 12345
    subs = []
    for k in range(outstanding):
        L, R = synth.make_batch(50 * k, B, W, H, nd)
        hl, hr = torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()
        hd = torch.full((B, H, W), sentinel, dtype=torch.int16).pin_memory()
        assert Lb.sbm_submit_dense(h, B, hl.data_ptr(), hr.data_ptr(), W, H, hd.data_ptr()) == 0
        subs.append((L, R, hl, hr, hd))
    Lb.sbm_destroy(h)                                  # nobody waited
    torch.cuda.synchronize()
    for k, (L, R, hl, hr, hd) in enumerate(subs):
        if k < outstanding - 1:                        # followed by another submission: its copy was queued and has finished
            assert np.array_equal(hd.numpy(), oracle.compute_batch(op, L, R)), k
        else:                                          # the newest: dropped, untouched
            assert (hd.numpy() == sentinel).all(), k
    # the parked handle is re-armed by the next create: nothing from the old life writes into the dropped destination
    bm = pkg.StereoBM.create(nd, 9)
    bm.setUniquenessRatio(10); bm.setDisp12MaxDiff(1)
    L, R = subs[-1][0], subs[-1][1]
    assert np.array_equal(bm.compute(L, R), oracle.compute_batch(op, L, R))
    assert (subs[-1][4].numpy() == sentinel).all()
    # ... and the mirror's close() drains: a submission followed by close() IS delivered
    out = torch.full((B, H, W), sentinel, dtype=torch.int16).pin_memory()
    bm.submit_host(subs[-1][2].numpy(), subs[-1][3].numpy(), out.numpy())
    bm.close()
    assert np.array_equal(out.numpy(), oracle.compute_batch(op, L, R))
