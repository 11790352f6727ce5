#!/usr/bin/env python3
"""bench.py -- throughput of the stereo block-matching hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole path (prefilter -> SAD/WTA -> LR check -> speckle) over one batch of synthetic
stereo pairs that is already resident in HBM. Workload at N=1: BASELINE.json configs[1] -- KITTI-shaped 1242x375 gray,
ndisp=128, 15x15 SAD, with the post-filter chain of the reference call site (src/slam/src/core/main.cpp:206-212:
texture 10, uniqueness 10, speckle 50/32, disp12MaxDiff 1) -- `--pairs` pairs per GPU per step (default 64 = the
per-GPU share of configs[3]). Multi-GPU: one process per GPU (torch.distributed / RCCL), pair batches sharded with no
data-path collective (weak scaling: per-GPU batch fixed).

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      dominant kernel (the SAD/WTA kernel): algorithmic HBM bytes per launch / its mean duration, measured
                with HIP events on the engine's stream inside the timed region (every 4th step is instrumented: the six
                event records of a step cost ~25 us); `traffic` = HBM bytes per launch from
                the committed rocprofv3 PMC run (profiles/), or null
  cpu_baseline  the CPU oracle (a port, not OpenCV) timed on this host's cores on a bounded sample of the same batch
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

WORKLOADS = {
    # name: (W, H, ndisp, block, default pairs per GPU)
    "kitti": (1242, 375, 128, 15, 64),       # BASELINE configs[1] / configs[3] per-GPU share
    "ref640": (640, 480, 64, 21, 64),        # configs[0] geometry with the reference's own window
    "fhd": (1920, 1080, 256, 21, 16),        # configs[2] geometry
    "uhd": (3840, 2160, 256, 21, 4),         # configs[4] geometry
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container can see
    256 CPUs and be limited to 16 CPUs' worth of time; threads beyond the quota only add throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]   # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            q = int(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())        # cgroup v1
            per = int(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return n


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_sg_child(rank, local_rank, world, port, argv, timeout_s):
    """The chunked RCCL scatter/gather leg in a FRESH child process of this rank (the children of all ranks form their own
    process group on `port`). The parent only waits: whatever the child does -- hang in a point-to-point batch, crash, never
    start -- costs at most `timeout_s` and can never touch the measurement the parent already holds. Returns rank 0's
    scatter_gather dictionary, or {"error": ...}."""
    import subprocess

    # (a launcher's own variables must not leak into the children: with TORCHELASTIC_USE_AGENT_STORE set, as torchrun does for its
    # workers, rank 0 would not host the rendezvous store on the new port and every child would wait for it until the limit)
    env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_ELASTIC_")) and k not in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")}
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(pathlib.Path(__file__).resolve())] + argv + ["--sg-child"]
    t0 = time.perf_counter()
    try:
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, text=True)
    except OSError as e:
        return {"error": f"could not start the scatter/gather child: {e}"}
    try:
        out, _ = child.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        child.kill()          # exactly the process this rank started
        child.communicate()
        return {"error": f"timeout: the scatter/gather child of rank {rank} did not finish within {timeout_s:.0f} s and was killed"}
    if child.returncode != 0:
        return {"error": f"the scatter/gather child of rank {rank} exited with status {child.returncode}"}
    if rank != 0:
        return None
    lines = [l for l in (out or "").splitlines() if l.startswith("{")]
    if not lines:
        return {"error": "the scatter/gather child of rank 0 printed no result"}
    sg = json.loads(lines[-1])
    sg["wall_s_incl_start_up"] = round(time.perf_counter() - t0, 2)
    return sg


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one child process per GPU (this process never touches the
    GPU, so nothing is exec'ed or forked after HIP initialisation), pass rank 0's JSON line through, fail if any rank
    fails. Equivalent to `python -m torch.distributed.run --nproc-per-node N bench.py ...`, which still works."""
    import subprocess

    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(pathlib.Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in procs:          # a dead rank leaves the others waiting in a collective: stop exactly those PIDs
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def probe_opencv():
    """SURVEY.md 8c / BASELINE.md 3.1: use the real cv::StereoBM as checker and CPU baseline if this host has it."""
    try:
        import cv2  # noqa: F401

        return cv2, f"cv2 {cv2.__version__}"
    except Exception as e:  # noqa: BLE001
        import ctypes.util

        lib = ctypes.util.find_library("opencv_calib3d")
        why = f"import cv2 failed ({type(e).__name__})"
        if lib:
            return None, f"unavailable: {why}; {lib} exists but cv::StereoBM has no C ABI to bind"
        return None, f"unavailable: {why}; no libopencv_calib3d on the loader path"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="kitti", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (0 = workload default)")
    ap.add_argument("--block", type=int, default=0, help="override the SAD window")
    ap.add_argument("--ndisp", type=int, default=0, help="override the number of disparities (experiments)")
    ap.add_argument("--no-postfilter", action="store_true", help="SAD/WTA/texture/uniqueness only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-profile", action="store_true", help="no stage events in the timed region (experiments)")
    ap.add_argument("--check", action="store_true", help="also verify pair 0 of rank 0 against the oracle")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: also gather every step's disparity maps on rank 0 (RCCL) inside the timed region")
    ap.add_argument("--scatter", action="store_true",
                    help="N>1: the whole global batch starts on rank 0; every step scatters it in chunks of --chunk pairs, "
                         "computes and gathers the maps back (double-buffered point-to-point over RCCL), all inside the timed region")
    ap.add_argument("--chunk", type=int, default=8, help="pairs per transfer chunk of --scatter")
    ap.add_argument("--feed", choices=("resident", "host"), default="resident",
                    help="host: every rank streams its own shard from PINNED host memory through the engine's chunked three-stream "
                         "host entry point (sbm_compute_batch), inputs and maps crossing PCIe inside the timed region")
    ap.add_argument("--prewarm-s", type=float, default=0.3,
                    help="untimed steps for at least this many seconds BEFORE the --warmup steps (clock ramp; disclosed as prewarm_s)")
    ap.add_argument("--no-sg", action="store_true", help="N>1: skip the scatter/gather leg")
    ap.add_argument("--no-multi", action="store_true", help="N>1: skip the one-process C++ leg (sbm_compute_batch_multi over the visible devices)")
    ap.add_argument("--sg-timeout", type=float, default=180.0, help="N>1: wall-clock limit of the scatter/gather child processes")
    ap.add_argument("--sg-child", action="store_true", help=argparse.SUPPRESS)   # internal: this process IS a scatter/gather child
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world   # an external launcher decides
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    # SBM_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a box with fewer GPUs than ranks (ranks then
    # share devices; numbers from such a run are meaningless). The driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("SBM_BENCH_BACKEND", "nccl")
    if backend != "nccl" or local_rank >= torch.cuda.device_count():
        # (also covers launchers that expose one device per process through HIP_VISIBLE_DEVICES)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import _pkg

    pkg = _pkg.load()
    from u96_slam_amd import synth

    W, H, nd, wsz, pairs_default = WORKLOADS[args.workload]
    if args.block:
        wsz = args.block
    if args.ndisp:
        nd = args.ndisp
    B = args.pairs or pairs_default
    post = not args.no_postfilter

    # ---- synthetic shard for this rank (distinct pairs per rank), resident in HBM before timing starts ----------
    uniq = min(B, 16)  # distinct pairs generated on the host; tiled up to B (keeps start-up short)
    Lh, Rh = synth.make_batch(rank * B, uniq, W, H, nd)
    reps = (B + uniq - 1) // uniq
    Lh = np.concatenate([Lh] * reps)[:B]
    Rh = np.concatenate([Rh] * reps)[:B]
    dev = torch.device("cuda", local_rank)
    dL, dR = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)
    dD = torch.empty((B, H, W), dtype=torch.int16, device=dev)
    scatter = args.scatter and world > 1
    gL = gR = gD = None
    xdev = dev if backend == "nccl" else torch.device("cpu")   # gloo moves host tensors
    if scatter and rank == 0:
        # the global batch (every rank's shard) resident on the root; pairs of rank r are generated as in the sharded run
        parts = [synth.make_batch(r * B, uniq, W, H, nd) for r in range(world)]
        gL = torch.from_numpy(np.concatenate([np.concatenate([pp[0]] * reps)[:B] for pp in parts])).to(xdev)
        gR = torch.from_numpy(np.concatenate([np.concatenate([pp[1]] * reps)[:B] for pp in parts])).to(xdev)
        gD = torch.empty((world * B, H, W), dtype=torch.int16, device=xdev)

    bm = pkg.StereoBM.create(nd, wsz, device=local_rank)
    bm.setPreFilterCap(31)
    bm.setMinDisparity(0)
    bm.setTextureThreshold(10)
    bm.setUniquenessRatio(10)
    if post:
        bm.setSpeckleWindowSize(50)
        bm.setSpeckleRange(32)
        bm.setDisp12MaxDiff(1)

    def sync_all():
        torch.cuda.synchronize(dev)
        bm.synchronize()
        feed_out[0] = 0
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    pl, pr, pd = dL.data_ptr(), dR.data_ptr(), dD.data_ptr()

    def compute_chunk(l, r):      # per-rank engine call of the chunked scatter path (finished when it returns)
        if l.is_cuda:
            return bm.compute_device(l, r, sync=True)
        return bm.compute_device(l.to(dev), r.to(dev), sync=True).cpu()

    host_feed = args.feed == "host" and not scatter
    hL = hR = hD = None
    if host_feed:
        # this rank's shard in pinned host memory (hipHostMalloc through torch): what a per-rank feeder thread would own
        hL, hR = torch.from_numpy(Lh).pin_memory(), torch.from_numpy(Rh).pin_memory()
        hD = torch.empty((B, H, W), dtype=torch.int16).pin_memory()
        hLn, hRn, hDn = hL.numpy(), hR.numpy(), hD.numpy()
        # second buffer set of the double-buffered feeder (batch k+1 is filled / submitted while batch k is in flight)
        feed_keep = [(hL.clone().pin_memory(), hR.clone().pin_memory(), torch.empty((B, H, W), dtype=torch.int16).pin_memory()) for _ in range(2)]
        feed_sets = [(hLn, hRn, hDn)] + [(a.numpy(), b.numpy(), c.numpy()) for a, b, c in feed_keep]
    feed_i, feed_out = [0], [0]   # submissions made / not yet waited for (sbm_synchronize drains the queue)
    feed_async = os.environ.get("SBM_BENCH_FEED", "async") == "async"   # SBM_BENCH_FEED=sync: one sbm_compute_batch call per step

    def step():
        if host_feed:
            if feed_async:
                # asynchronous dense feed (sbm_submit_dense / sbm_wait_oldest): batch k+1 crosses PCIe while batch k computes
                # and batch k-1's maps return; the region's closing sbm_synchronize() waits for the last submissions
                bm.submit_host(*feed_sets[feed_i[0] % 3])
                feed_i[0] += 1
                feed_out[0] += 1
                if feed_out[0] >= 3:         # three batches in flight: one arriving, one computing, one leaving
                    bm.wait_host()
                    feed_out[0] -= 1
                return
            # one synchronous call of the host entry point (sbm_compute_batch): inside the call chunks of pairs flow through an
            # H2D stream, the compute stream and a D2H stream; it returns when the maps are in the caller's (pinned) memory
            bm.compute(hLn, hRn, hDn)
            return
        if scatter:
            from u96_slam_amd import shard

            shard.compute_sharded_chunked(compute_chunk, gL, gR, world * B, (H, W), chunk=args.chunk, src=0, device=xdev, out=gD)
        else:
            bm.launch_raw(B, pl, pr, W, H, pd)

    # ---- the scatter/gather leg (runs in child processes of the timed ranks, see run_sg_child) --------------------------
    def sg_leg():
        from u96_slam_amd import shard

        fault = os.environ.get("SBM_BENCH_SG_FAULT", "")      # tests: "hang" / "crash" on the last rank
        if fault and rank == world - 1:
            if fault == "hang":
                time.sleep(3600)
            os._exit(7)
        nsg = max(1, min(args.steps, 10))
        sL = sR = sD = None
        if rank == 0:
            parts = [synth.make_batch(r * B, uniq, W, H, nd) for r in range(world)]
            sL = torch.from_numpy(np.concatenate([np.concatenate([pp[0]] * reps)[:B] for pp in parts])).to(xdev)
            sR = torch.from_numpy(np.concatenate([np.concatenate([pp[1]] * reps)[:B] for pp in parts])).to(xdev)
            sD = torch.empty((world * B, H, W), dtype=torch.int16, device=xdev)

        def sg_step():
            shard.compute_sharded_chunked(compute_chunk, sL, sR, world * B, (H, W), chunk=args.chunk, src=0, device=xdev, out=sD)

        sg_step()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(nsg):
            sg_step()
        sync_all()
        tsg = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tsg, op=dist.ReduceOp.MAX)
        sg_ms = float(tsg.item()) / nsg * 1e3
        # the same chunks computed back to back without any transfer: what the scatter/gather adds is sg_ms - that
        t1 = time.perf_counter()
        for _ in range(nsg):
            for c0 in range(0, B, args.chunk):
                bm.compute_device(dL[c0:c0 + args.chunk], dR[c0:c0 + args.chunk], sync=True)
        tcc = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tcc, op=dist.ReduceOp.MAX)
        chunk_ms = float(tcc.item()) / nsg * 1e3
        in_bytes, out_bytes = 2.0 * (world - 1) * B * W * H, 2.0 * (world - 1) * B * W * H
        return {"ms_per_step": round(sg_ms, 4), "value": round(world * B * W * H * nd / (sg_ms * 1e-3) / 1e6, 2),
                "unit": "Mpix-disparities/s", "steps": nsg, "chunk": args.chunk, "backend": "RCCL" if backend == "nccl" else backend,
                "compute_only_ms_per_step": round(chunk_ms, 4),
                "root_link_GBps": round((in_bytes + out_bytes) / (sg_ms * 1e-3) / 1e9, 2),
                "overlap_frac": round(max(0.0, min(1.0, chunk_ms / sg_ms)), 4) if sg_ms > 0 else None,
                "note": "measured in fresh child processes of the timed ranks (own process group, wall-clock limit): global batch "
                        "resident on rank 0, chunked double-buffered point-to-point scatter of the pairs + gather of the maps inside "
                        "the timed region; overlap_frac = per-rank chunked compute time / scatter-gather step time (1.0 = the "
                        "transfers are completely hidden under the computation)"}

    if args.sg_child:
        if dist is None:
            sys.exit("--sg-child is internal to bench.py --gpus N")
        res = sg_leg()
        if rank == 0:
            print(json.dumps(res), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # untimed: a time-based pre-warm (the clocks need some hundred milliseconds of load to settle; --warmup 5 is 5 ms), then
    # the --warmup steps the command line asks for
    prewarm_steps = 0
    if args.prewarm_s > 0 and not scatter:      # (a scattered step holds collectives: every rank must run the same number)
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < args.prewarm_s:
            for _ in range(8):
                step()
            prewarm_steps += 8
            bm.synchronize()
            feed_out[0] = 0
    for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
        step()
    sync_all()

    # stage events recorded on the engine's stream inside the timed region, no host sync; every 4th step is instrumented (six
    # event records cost ~25 us per step: sampling keeps the timed rate within ~0.5 % of the un-instrumented one)
    bm.set_profiling(0 if (args.no_profile or host_feed) else (3 if args.steps >= 8 else 2))
    sync_all()
    # two event records on the engine's stream bracket the first five timed steps (clock ramp made visible, not hidden)
    first5 = None
    mark5 = not scatter and not host_feed and args.steps > 5
    if mark5:
        es5 = torch.cuda.ExternalStream(bm.stream(), device=dev)
        ev5 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    gathered = None
    if mark5:
        ev5[0].record(es5)
    for istep in range(args.steps):
        if mark5 and istep == 5:
            ev5[1].record(es5)
        step()
        if args.gather and dist is not None and not scatter:
            from u96_slam_amd import shard

            bm.synchronize()
            src = dD if backend == "nccl" else dD.cpu()
            gathered = shard.gather_disparities(src, world * B, dst=0)
    if mark5:
        ev5[2].record(es5)
    bm.synchronize()
    feed_out[0] = 0
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if mark5:
        first5 = (ev5[0].elapsed_time(ev5[1]) / 5.0, ev5[1].elapsed_time(ev5[2]) / (args.steps - 5))
    prof = bm.profile()
    bm.set_profiling(0)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_name = bm.last_kernel()

    # ---- per-step distribution (SURVEY.md 8d "report median and min"): a further pass of min(steps, 50) steps with one HIP event
    # per step on the ENGINE's stream (torch.cuda.Event only sees torch's streams unless it is recorded on this external one);
    # outside the timed region, so `value` is not perturbed
    step_ms = []
    if not scatter:
        nper = max(2, min(args.steps, 50))
        if host_feed:
            bm.synchronize()
            feed_out[0] = 0
            for _ in range(nper):           # host time between consecutive completions (the queue stays three deep)
                t1 = time.perf_counter()
                step()
                step_ms.append((time.perf_counter() - t1) * 1e3)
            bm.synchronize()
            feed_out[0] = 0
            step_ms = step_ms[3:] if len(step_ms) > 6 else step_ms   # (the first submissions only fill the queue)
        else:
            es = torch.cuda.ExternalStream(bm.stream(), device=dev)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nper + 1)]
            evs[0].record(es)
            for i in range(nper):
                step()
                evs[i + 1].record(es)
            bm.synchronize()
            torch.cuda.synchronize(dev)
            step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(nper)]
    step_ms.sort()

    # ---- multi-GPU evidence in the default line (no flag needed): rank identity over the real backend, and min(steps, 10)
    # steps of the chunked double-buffered point-to-point scatter/gather from rank 0 (the path north_star names)
    rccl = sg = None
    if dist is not None:
        props = torch.cuda.get_device_properties(local_rank)
        ident = str(getattr(props, "uuid", "")) or f"{getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}"
        idents = [None] * world
        dist.all_gather_object(idents, (rank, local_rank, ident))
        rccl = {"backend": backend, "ranks_seen": len(idents), "distinct_devices": len({i[2] for i in idents}),
                "devices": [i[2] for i in idents]}
        if not scatter and not host_feed and not args.no_sg:
            # the point-to-point leg runs in fresh child processes with a wall-clock limit: a hang or fault in there costs
            # sg_timeout seconds and an error field, never the measurement above
            port = [free_port() if rank == 0 else None]
            dist.broadcast_object_list(port, src=0)
            dist.barrier()
            sg = run_sg_child(rank, int(os.environ.get("LOCAL_RANK", "0")), world, port[0], sys.argv[1:], args.sg_timeout)

    # ---- the reference's own shape of "several GPUs": ONE process, one host thread, one engine per visible device and the whole
    # global batch in one call (sbm_compute_batch_multi, INTEGRATION.md "Several GPUs from one C++ process"). Rank 0 runs it
    # while the other ranks wait at a barrier; PCIe is inside (pinned host batch), so this is a host-feed rate, never `value`.
    multi = None
    if dist is not None and not scatter and not host_feed and not args.no_multi and not args.sg_child:
        sync_all()
        if rank == 0:
            try:
                ndev = torch.cuda.device_count()
                n_glob = max(ndev, min(world * B, int(1.5e9 // (4 * W * H))))      # at most ~1.5 GB of pinned host memory
                engines = []
                for k in range(ndev):
                    e = pkg.StereoBM.create(nd, wsz, device=k)
                    e.setPreFilterCap(31); e.setMinDisparity(0); e.setTextureThreshold(10); e.setUniquenessRatio(10)
                    if post:
                        e.setSpeckleWindowSize(50); e.setSpeckleRange(32); e.setDisp12MaxDiff(1)
                    engines.append(e)
                mreps = (n_glob + uniq - 1) // uniq
                mL = torch.from_numpy(np.concatenate([Lh[:uniq]] * mreps)[:n_glob]).pin_memory()
                mR = torch.from_numpy(np.concatenate([Rh[:uniq]] * mreps)[:n_glob]).pin_memory()
                mD = torch.empty((n_glob, H, W), dtype=torch.int16).pin_memory()
                pkg.compute_multi(engines, mL.numpy(), mR.numpy(), mD.numpy())       # warm: scratch, staging sets, streams
                nrep = max(2, min(args.steps, 5))
                t1 = time.perf_counter()
                for _ in range(nrep):
                    pkg.compute_multi(engines, mL.numpy(), mR.numpy(), mD.numpy())
                mms = (time.perf_counter() - t1) / nrep * 1e3
                same = bool(torch.equal(mD[:min(B, n_glob)], dD[:min(B, n_glob)].cpu()))   # block 0 = this rank's resident shard
                multi = {"api": "sbm_compute_batch_multi", "devices": ndev, "pairs_per_call": n_glob, "ms_per_call": round(mms, 4),
                         "value": round(n_glob * W * H * nd / (mms * 1e-3) / 1e6, 2), "unit": "Mpix-disparities/s",
                         "equals_resident_shard": same,
                         "note": "one process, one host thread, one engine per visible device, contiguous pair blocks; pinned host "
                                 "batch, H2D + compute + D2H per device inside the call (PCIe-inclusive: not comparable with value)"}
                del engines
            except Exception as ex:      # this leg must never cost the line
                multi = {"error": f"{type(ex).__name__}: {ex}"}
        sync_all()

    # ---- host feed: the three legs on their own (resident compute, H2D of the inputs, D2H of the maps) -> how much of the
    # shorter legs the three-stream pipeline hides: overlap_frac = 1 when a step costs only its slowest leg, 0 when the sum
    feed_legs = None
    if host_feed:
        nleg = max(2, min(args.steps, 20))
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(nleg):
            dL.copy_(hL, non_blocking=True)
            dR.copy_(hR, non_blocking=True)
        e1.record()
        for _ in range(nleg):
            hD.copy_(dD, non_blocking=True)
        e2.record()
        torch.cuda.synchronize(dev)
        h2d_ms, d2h_ms = e0.elapsed_time(e1) / nleg, e1.elapsed_time(e2) / nleg
        bm.launch_raw(B, pl, pr, W, H, pd)     # (re-sizes the engine's scratch from chunk to batch size: not timed)
        bm.synchronize()
        t1 = time.perf_counter()
        for _ in range(nleg):
            bm.launch_raw(B, pl, pr, W, H, pd)
        bm.synchronize()
        res_ms = (time.perf_counter() - t1) / nleg * 1e3
        # the synchronous entry point on the same data: one sbm_compute_batch call per step (chunked three-stream pipeline
        # INSIDE the call; nothing of the next call can start before this one has returned)
        nq = max(3, min(args.steps, 20))
        bm.compute(hLn, hRn, hDn)
        t1 = time.perf_counter()
        for i in range(nq):
            bm.compute(hLn, hRn, hDn)
        queue_ms = (time.perf_counter() - t1) / nq * 1e3
        feed_ms = elapsed / args.steps * 1e3
        legs = (res_ms, h2d_ms, d2h_ms)
        feed_legs = {"resident_compute_ms": round(res_ms, 4), "h2d_ms": round(h2d_ms, 4), "d2h_ms": round(d2h_ms, 4),
                     "h2d_alone_GBps": round(2.0 * B * W * H / (h2d_ms * 1e-3) / 1e9, 2), "d2h_alone_GBps": round(2.0 * B * W * H / (d2h_ms * 1e-3) / 1e9, 2),
                     "overlap_frac": round(max(0.0, min(1.0, 1.0 - (feed_ms - max(legs)) / max(sum(legs) - max(legs), 1e-9))), 4),
                     "sync_call_ms_per_step": round(queue_ms, 4)}

    total_pairs = world * B * args.steps
    pixdisp_per_pair = W * H * nd
    value = total_pairs * pixdisp_per_pair / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        # ---- roofline of the dominant kernel ---------------------------------------------------------------------
        kms = prof["sad"]   # the SAD/WTA kernel of the call (interior kernel -- beyond 256 disparities with its two border launches -- or a fallback one: `kernel` says which)
        algo_bytes = 4.0 * W * H * B  # read L+R (2 B/px) + write int16 disparity (2 B/px) per pair
        achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        traffic = None
        pmc_extra = {}
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                key = f"{args.workload}_w{wsz}_b{B}"
                sys.path.insert(0, str(ROOT / "tools"))
                import publish_profiles

                here = publish_profiles.source_hash(ROOT)
                if key in tj and tj[key].get("source_hash") and tj[key]["source_hash"] != here:
                    # same kernel name, other sources: counters of another build are never attached to this one
                    pmc_extra = {"traffic_reason": f"profiles/hbm_traffic.json[{key}] was measured on engine sources {tj[key]['source_hash']}, this tree "
                                                   f"is {here}: re-run tools/profile_round.sh"}
                elif key in tj and tj[key].get("kernel") and tj[key]["kernel"] != kernel_name:
                    # the committed counters were taken on another kernel instantiation: never attach them to this one
                    pmc_extra = {"traffic_reason": f"profiles/hbm_traffic.json[{key}] was measured on '{tj[key]['kernel']}', this run launched "
                                                   f"'{kernel_name}': re-run tools/profile_round.sh"}
                elif key in tj:
                    traffic = tj[key].get("bytes_per_launch")
                    # what actually bounds the kernel: VALU issue, from the same committed PMC run
                    pmc_extra = {k: tj[key][k] for k in ("valu_busy_frac", "lds_busy_frac", "lane_ops_per_pixel_disparity") if k in tj[key]}
                    pmc_extra["traffic_from"] = f"{tj[key].get('source')} ({tj[key].get('date', 'round 1')}, commit {tj[key].get('commit', '?')}), not this run"
            except Exception:
                traffic = None
        # `bound` follows the bench contract ("hbm" | "mfma"): the HBM figure is what north_star asks for; the kernel itself
        # is limited by VALU issue (`limiter`, `valu_busy_frac`), so `frac` is informational, not a measure of its quality
        roofline = {"bound": "hbm", "limiter": "valu", "kernel": kernel_name,
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel_ms": round(kms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                    "stage_ms": {k: round(v, 4) for k, v in prof.items()},
                    "stage_ms_from": "HIP events on every 4th timed step (an instrumented step runs ~20 us longer than the others)"
                    if (not args.no_profile and args.steps >= 8) else "HIP events on every timed step", **pmc_extra}

        # the roof that does bound the kernel (SURVEY.md 8d "algorithmic ops"): vector lane-operations per second. Peak = 256 CUs x
        # 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s with every instruction at full rate; the kernel's v_mqsad_pk_u16_u8
        # runs at quarter rate, which is why `busy_frac` (VALU busy cycles, from the committed counter run) exceeds `frac`
        if "lane_ops_per_pixel_disparity" in pmc_extra and kms > 0:
            lane_ops = pmc_extra["lane_ops_per_pixel_disparity"] * B * W * H * nd
            peak_lane = 256 * 4 * 16 * 2.4e9
            # (which peak this is: the 4-cycle issue model -- one wave64 instruction per SIMD and 4 cycles. MI355X_MICROARCH.md
            # lists SIMD-32 units with v_fma_f32 at 2 cycles per wave64, i.e. 78.6 T lane-ops/s for full-rate instructions, and this
            # repo's own probe has add / sub / logic at 2.5-2.9 cycles, everything else at 4.3-4.9, v_mqsad_pk_u16_u8 at 17: `frac`
            # is against the model, `frac_full_rate` against the guide's figure; `busy_frac` -- the counter -- is the evidence)
            peak_full = 256 * 4 * 32 * 2.4e9
            roofline["valu"] = {"achieved": round(lane_ops / (kms * 1e-3) / 1e12, 2), "peak": round(peak_lane / 1e12, 1),
                                "unit": "Tlane-op/s", "frac": round(lane_ops / (kms * 1e-3) / peak_lane, 4),
                                "peak_model": "4-cycle issue, 16 lanes/SIMD/clk (256 CUs x 4 SIMDs x 16 x 2.4 GHz)",
                                "peak_full_rate": round(peak_full / 1e12, 1), "frac_full_rate": round(lane_ops / (kms * 1e-3) / peak_full, 4),
                                "peak_full_rate_model": "SIMD-32, full-rate instructions only (MI355X_MICROARCH.md: v_fma_f32 2 cycles per wave64)",
                                "busy_frac": pmc_extra.get("valu_busy_frac")}

        # the one stage of the path that IS HBM-bound: the prefilter, 1 B read + 1 B written per pixel and image
        pf_ms = prof.get("prefilter", 0.0)
        pf_bytes = 4.0 * W * H * B
        # the device's own copy rate for the same bytes (read 2*W*H*B, write the same) with the runtime's copy kernel, cycling
        # through enough buffers (> 512 MB) that the copies come from HBM like the prefilter's inputs do inside a step --
        # a single 119 MB buffer pair would sit in the 256 MB Infinity Cache and flatter the reference
        nbuf = max(2, int(-(-(768 << 20) // max(1, 4 * B * H * W))))
        cp_src = [torch.empty((2, B, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        cp_dst = [torch.empty((2, B, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        for i in range(nbuf):
            cp_dst[i].copy_(cp_src[i])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ncopy = 3 * nbuf
        e0.record()
        for i in range(ncopy):
            cp_dst[i % nbuf].copy_(cp_src[i % nbuf])
        e1.record()
        torch.cuda.synchronize(dev)
        copy_ms = e0.elapsed_time(e1) / ncopy
        copy_gbs = pf_bytes / (copy_ms * 1e-3) / 1e9 if copy_ms > 0 else 0.0
        del cp_src, cp_dst
        roofline_pf = {"bound": "hbm", "kernel": "prefilter_kernel", "achieved": round(pf_bytes / (pf_ms * 1e-3) / 1e9, 2) if pf_ms > 0 else 0.0,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pf_bytes / (pf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pf_ms > 0 else 0.0,
                       "kernel_ms": round(pf_ms, 4), "algorithmic_bytes_per_launch": pf_bytes,
                       "device_copy_GBps": round(copy_gbs, 1), "frac_of_device_copy": round(pf_bytes / (pf_ms * 1e-3) / 1e9 / copy_gbs, 4) if pf_ms > 0 and copy_gbs > 0 else 0.0,
                       "note": "kernel_ms is the event-to-event stage time inside the engine (includes ~5 us of launch gap); stand-alone "
                               "kernel rates inside and beyond the Infinity Cache: profiles/r02_prefilter.json"}

        # ---- CPU baseline (reported only) --------------------------------------------------------------------------
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
            import sbm_oracle

            cores = max(1, min(usable_cores(), sbm_oracle.max_threads()))
            p = sbm_oracle.make_params(nd, wsz, 31, 0, 10, 10, 50 if post else 0, 32 if post else 0, 1 if post else -1)
            cv2, cv_status = probe_opencv()
            if cv2 is not None:
                # the real cv::StereoBM of this host: checker for the GPU output and CPU baseline
                cv2.setNumThreads(cores)
                m = cv2.StereoBM_create(numDisparities=nd, blockSize=wsz)
                m.setPreFilterCap(31); m.setMinDisparity(0); m.setTextureThreshold(10); m.setUniquenessRatio(10)
                if post:
                    m.setSpeckleWindowSize(50); m.setSpeckleRange(32); m.setDisp12MaxDiff(1)
                t1 = time.perf_counter()
                m.compute(Lh[0], Rh[0])
                one = time.perf_counter() - t1
                sample = args.cpu_sample or max(4, min(B, int(round(15.0 / max(one, 1e-3)))))
                got = dD[:min(sample, B)].cpu().numpy()
                t1 = time.perf_counter()
                ref = [m.compute(Lh[i % B], Rh[i % B]) for i in range(sample)]
                cpu_s = time.perf_counter() - t1
                exact = all(np.array_equal(got[i], ref[i]) for i in range(min(sample, B)))
                cpu = {"value": round(sample * pixdisp_per_pair / cpu_s / 1e6, 2), "unit": "Mpix-disparities/s", "cores": cores,
                       "kind": "reference", "opencv": cv_status, "bit_exact_vs_gpu": bool(exact),
                       "sample": f"{sample} pairs of the same {W}x{H} nd{nd} w{wsz} batch, {cpu_s:.2f} s wall, cv2.StereoBM with cv2.setNumThreads({cores})"}
            else:
                t1 = time.perf_counter()
                sbm_oracle.compute_batch(p, Lh[:1], Rh[:1], threads=1)
                one = time.perf_counter() - t1
                # about 20 s of single-core work in total, at least one pair per core
                sample = args.cpu_sample or max(cores, min(8 * cores, int(round(20.0 / max(one, 1e-3)))))
                idx = [i % B for i in range(sample)]
                t1 = time.perf_counter()
                ref = sbm_oracle.compute_batch(p, Lh[idx], Rh[idx], threads=cores)
                cpu_s = time.perf_counter() - t1
                cpu = {"value": round(sample * pixdisp_per_pair / cpu_s / 1e6, 2), "unit": "Mpix-disparities/s", "cores": cores,
                       "kind": "port", "opencv": cv_status,
                       "sample": f"{sample} pairs of the same {W}x{H} nd{nd} w{wsz} batch, {cpu_s:.2f} s wall, "
                       "in-repo C restatement of cv::StereoBM (not OpenCV), OpenMP across pairs"}
                # the same sample through the u16-vectorised correspondence stage (oracle/sbm_oracle_simd.c: 16-bit sums, AVX2 by
                # auto-vectorisation -- the shape of cv::StereoBM's own "useShorts" SIMD path, which the scalar int32 port above
                # understates). Checked equal to the scalar port; `value` becomes the faster, more representative figure and the
                # scalar one stays beside it, each under its own label.
                if sbm_oracle.simd_ok(p):
                    with sbm_oracle.simd():
                        sbm_oracle.compute_batch(p, Lh[:cores], Rh[:cores], threads=cores)       # (first touch of its buffers)
                        t1 = time.perf_counter()
                        ref_v = sbm_oracle.compute_batch(p, Lh[idx], Rh[idx], threads=cores)
                        simd_s = time.perf_counter() - t1
                    cpu["scalar_value"] = cpu["value"]
                    cpu["value"] = round(sample * pixdisp_per_pair / simd_s / 1e6, 2)
                    cpu["variant"] = "u16-vectorised correspondence stage (16 x u16 per AVX2 register, gcc auto-vectorisation)"
                    cpu["simd_equals_scalar"] = bool(np.array_equal(ref_v, ref))
                    cpu["sample"] = (f"{sample} pairs of the same {W}x{H} nd{nd} w{wsz} batch: u16-vectorised port {simd_s:.2f} s wall (value), "
                                     f"scalar int32 port {cpu_s:.2f} s wall (scalar_value); in-repo C restatements of cv::StereoBM "
                                     "(not OpenCV), OpenMP across pairs")
                if args.check:
                    got = dD[: min(sample, B)].cpu().numpy()
                    ok = all(np.array_equal(got[i], ref[i]) for i in range(min(sample, B)) if idx[i] == i)
                    cpu["bit_exact_vs_gpu"] = bool(ok)

        out = {
            "metric": "Mpix-disparities/s", "value": round(value, 2), "unit": "Mpix-disparities/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": f"synthetic ({uniq} distinct KITTI-style pairs per rank tiled to {B}; SURVEY.md 8d generator)",
            "config": {"workload": f"{args.workload}: {W}x{H} gray, ndisp={nd}, {wsz}x{wsz} SAD, "
                       + ("texture 10 / uniqueness 10 / disp12MaxDiff 1 / speckle 50,32" if post else "texture 10 / uniqueness 10, no LR/speckle"),
                       "pairs_per_gpu_per_step": B, "global_pairs_per_step": world * B, "parallelism": f"pairs sharded x{world}, " + (
                           f"global batch on rank 0: chunked ({args.chunk} pairs) double-buffered point-to-point scatter + gather over {'RCCL' if backend == 'nccl' else backend} inside the timed region" if scatter
                           else "disparity maps gathered on rank 0 each step" if (args.gather and world > 1)
                           else "every rank streams its shard from pinned host memory (PCIe inside the timed region)" if host_feed
                           else "shards resident, no data-path collective (value); scatter_gather = the chunked RCCL scatter/gather from rank 0")},
            "prewarm_s": args.prewarm_s, "prewarm_steps": prewarm_steps,
            "ms_per_step_first5": round(first5[0], 4) if first5 else None,
            "ms_per_step_after5": round(first5[1], 4) if first5 else None,
            "ms_per_step_median": round(step_ms[len(step_ms) // 2], 4) if step_ms else None,
            "ms_per_step_min": round(step_ms[0], 4) if step_ms else None,
            "ms_per_pair": round(elapsed / (B * args.steps) * 1e3, 5),
            "pairs_per_s": round(total_pairs / elapsed, 1),
            "roofline": roofline, "roofline_prefilter": roofline_pf, "cpu_baseline": cpu,
            "engine_library": pkg.stereobm.loaded_library_name(),
        }
        if step_ms:
            out["ms_per_step_from"] = (f"{len(step_ms)} further steps, " + ("host wall clock per sbm_compute_batch call" if host_feed else "one HIP event per step on the engine's stream"))
        if rccl is not None:
            out["rccl"] = rccl
        if sg is not None:
            out["scatter_gather"] = sg
        if multi is not None:
            out["multi_cxx"] = multi
        if host_feed:
            nbytes_in, nbytes_out = 2.0 * B * W * H, 2.0 * B * W * H
            out["host_feed"] = {"memory": "pinned (torch pin_memory = hipHostMalloc)", "h2d_GBps": round(nbytes_in / (ms_per_step * 1e-3) / 1e9, 2),
                                "d2h_GBps": round(nbytes_out / (ms_per_step * 1e-3) / 1e9, 2),
                                "bytes_per_step_each_way": nbytes_in, **(feed_legs or {}),
                                "api": "sbm_submit_dense / sbm_wait_oldest" if feed_async else "sbm_compute_batch",
                                "note": "every rank feeds its own shard from pinned host memory through the asynchronous dense feed (whole batches, "
                                        "three in flight: H2D stream | compute stream | D2H stream); rates are bytes per direction over the whole "
                                        "step; overlap_frac = 1 when a step costs only its slowest leg (measured alone: resident_compute_ms, h2d_ms, "
                                        "d2h_ms), 0 when it costs their sum; sync_call_ms_per_step = the same batch through one synchronous "
                                        "sbm_compute_batch call per step (chunked pipeline inside the call)"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()     # (no barrier: the ranks have nothing left to agree on)


if __name__ == "__main__":
    main()
