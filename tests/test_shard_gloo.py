"""N>1 path on CPU: gloo processes (world size 2, 3, 4 and 8) exercise the scatter -> per-rank compute -> gather sharding,
whole-block and chunked / double-buffered, with the CPU oracle standing in for the per-rank engine call (the engine itself
needs a GPU)."""
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


def _worker(rank, world, port, cases, q):
    """One gloo rank: runs every (n_pairs, chunk) case of `cases` in the same process group (chunk 0 = whole blocks)."""
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
    calls = []

    def compute_fn(l, r):   # stand-in for StereoBM.compute on a GPU rank
        calls.append(l.shape[0])
        return torch.from_numpy(sbm_oracle.compute_batch(p, l.numpy(), r.numpy(), threads=1))

    for ci, (n_pairs, chunk) in enumerate(cases):
        del calls[:]
        L = R = None
        if rank == 0:
            Ln, Rn = synth.make_batch(7 * ci, n_pairs, w, h, nd)
            L, R = torch.from_numpy(Ln), torch.from_numpy(Rn)
        if chunk:
            out = shard.compute_sharded_chunked(compute_fn, L, R, n_pairs, (h, w), chunk=chunk, src=0)
        else:
            out = shard.compute_sharded(compute_fn, L, R, n_pairs, (h, w), src=0)
        lo, hi = shard.shard_bounds(n_pairs, rank, world)
        # every rank computed exactly its block, in pieces of at most `chunk` pairs
        assert sum(calls) == hi - lo and (not chunk or all(0 < c <= chunk for c in calls)), (rank, n_pairs, chunk, calls)
        if rank == 0:
            ref = sbm_oracle.compute_batch(p, Ln, Rn, threads=1)
            q.put(("result", ci, bool(np.array_equal(out.numpy(), ref)), tuple(out.shape)))
        else:
            assert out is None
        q.put(("bounds", ci, rank, lo, hi))
        dist.barrier()
    dist.destroy_process_group()


def _run_world(world, cases, join_s=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(join_s)
            assert p.exitcode == 0, f"rank process ended with {p.exitcode}"
    finally:
        for p in procs:      # a failed or hung rank leaves the others in a receive: stop exactly those processes
            if p.is_alive():
                p.kill()
    msgs = [q.get(timeout=5) for _ in range(len(cases) * (world + 1))]
    for ci, (n_pairs, chunk) in enumerate(cases):
        res = [m for m in msgs if m[0] == "result" and m[1] == ci]
        assert len(res) == 1 and res[0][2] is True and res[0][3][0] == n_pairs, (world, n_pairs, chunk, res)
        bounds = sorted(m[2:] for m in msgs if m[0] == "bounds" and m[1] == ci)
        assert [b[0] for b in bounds] == list(range(world)) and bounds[0][1] == 0 and bounds[-1][2] == n_pairs
        assert all(bounds[i][2] == bounds[i + 1][1] for i in range(world - 1))


@pytest.mark.parametrize("n_pairs,chunk", [(4, 0), (5, 0), (1, 0),     # whole blocks, point to point
                                           (11, 2), (8, 8), (7, 3), (1, 2)])  # chunked + double-buffered, ragged last chunks
def test_scatter_compute_gather_world2(n_pairs, chunk):
    _run_world(2, [(n_pairs, chunk)], join_s=120)


@pytest.mark.parametrize("world", [3, 4, 8])
def test_scatter_compute_gather_more_ranks(world):
    """The rank-count-dependent paths of the chunked, three-deep pipeline (SURVEY.md 8e asks for 8 ranks): fewer pairs than
    ranks (ranks without a block post nothing and must not stall the others), ranks with fewer chunks than the longest
    block has steps, a chunk larger than any block, ragged last chunks, and the whole-block path with empty ranks --
    all cases of one world size run in one process group, back to back."""
    cases = [
        (1, 0), (world - 1, 0), (world + 1, 0),                 # whole blocks: empty ranks, ragged blocks
        (1, 2), (world - 1, 1), (world, 1),                     # chunked: n_pairs < world, exactly one pair each
        (2 * world + 1, 1),                                     # rank 0 has 3 chunks, the others 2 (uneven step counts)
        (world + 2, 8),                                         # chunk larger than every block
        (5 * world + 3, 2),                                     # 3 ranks with 6 pairs (3 chunks), the rest 5 (ragged third chunk)
        (3 * world, 3),                                         # one chunk per rank == block size
    ]
    _run_world(world, cases)


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
