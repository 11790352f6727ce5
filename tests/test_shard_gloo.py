"""N>1 path on CPU: world_size-2 gloo processes exercise the scatter -> per-rank compute -> gather sharding with the
CPU oracle standing in for the per-rank engine call (the engine itself needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q, chunk=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _pkg
    import sbm_oracle

    _pkg.load()
    from u96_slam_amd import shard, synth

    h, w, nd = 40, 96, 16
    p = sbm_oracle.make_params(nd, 9, 31, 0, 10, 10, 10, 16, 1)

    def compute_fn(l, r):   # stand-in for StereoBM.compute on a GPU rank
        return torch.from_numpy(sbm_oracle.compute_batch(p, l.numpy(), r.numpy(), threads=1))

    L = R = None
    if rank == 0:
        Ln, Rn = synth.make_batch(0, n_pairs, w, h, nd)
        L, R = torch.from_numpy(Ln), torch.from_numpy(Rn)
    if chunk:
        out = shard.compute_sharded_chunked(compute_fn, L, R, n_pairs, (h, w), chunk=chunk, src=0)
    else:
        out = shard.compute_sharded(compute_fn, L, R, n_pairs, (h, w), src=0)
    lo, hi = shard.shard_bounds(n_pairs, rank, world)
    if rank == 0:
        ref = sbm_oracle.compute_batch(p, Ln, Rn, threads=1)
        q.put(("result", bool(np.array_equal(out.numpy(), ref)), out.shape))
    else:
        assert out is None
    q.put(("bounds", rank, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs,chunk", [(4, 0), (5, 0), (1, 0),     # whole blocks, point to point
                                           (11, 2), (8, 8), (7, 3), (1, 2)])  # chunked + double-buffered, ragged last chunks
def test_scatter_compute_gather_world2(n_pairs, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q, chunk)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    msgs = [q.get(timeout=5) for _ in range(3)]
    res = [m for m in msgs if m[0] == "result"][0]
    assert res[1] is True and res[2][0] == n_pairs
    bounds = sorted(m[1:] for m in msgs if m[0] == "bounds")
    assert bounds[0][1] == 0 and bounds[0][2] == bounds[1][1] and bounds[1][2] == n_pairs


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    import _pkg

    _pkg.load()
    from u96_slam_amd import shard

    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
