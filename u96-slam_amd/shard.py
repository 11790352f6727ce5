"""One-process-per-GPU sharding of stereo-pair batches.

Pairs are independent (the reference even rebuilds its matcher per frame, src/slam/src/core/main.cpp:201), so a batch
of N pairs is split into contiguous blocks, one per rank, and the only communication is moving inputs out and
disparity maps back: torch.distributed scatter / gather (RCCL over xGMI with backend "nccl", gloo on CPU). There is
no collective inside the compute path. `compute_fn` is the per-rank engine call (StereoBM.compute on a GPU rank).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_pairs, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; the first n_pairs % world ranks get one extra pair."""
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_pairs, world):
    return [shard_bounds(n_pairs, r, world)[1] - shard_bounds(n_pairs, r, world)[0] for r in range(world)]


def scatter_pairs(left, right, n_pairs, shape_hw, src=0, device=None, group=None):
    """Rank `src` holds left/right (N,H,W) uint8; every rank returns its (n_r,H,W) block. Ragged blocks are padded
    to the largest block for the collective and trimmed afterwards."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes = shard_sizes(n_pairs, world)
    pad = max(sizes)
    h, w = shape_hw
    if device is None:
        device = left.device if left is not None else torch.device("cpu")
    outs = []
    for full in (left, right):
        recv = torch.empty((pad, h, w), dtype=torch.uint8, device=device)
        chunks = None
        if rank == src:
            chunks = []
            for r in range(world):
                lo, hi = shard_bounds(n_pairs, r, world)
                c = torch.zeros((pad, h, w), dtype=torch.uint8, device=device)
                c[: hi - lo] = full[lo:hi].to(device)
                chunks.append(c)
        dist.scatter(recv, chunks, src=src, group=group)
        outs.append(recv[: sizes[rank]].contiguous())
    return outs[0], outs[1]


def gather_disparities(local_disp, n_pairs, dst=0, group=None):
    """Inverse of scatter_pairs for the int16 maps. Returns (N,H,W) on rank `dst`, None elsewhere."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes = shard_sizes(n_pairs, world)
    pad = max(sizes)
    h, w = local_disp.shape[-2:]
    send = torch.zeros((pad, h, w), dtype=torch.int16, device=local_disp.device)
    send[: sizes[rank]] = local_disp
    # transported as raw bytes: every backend moves uint8 (gloo has no int16 gather)
    send8 = send.view(torch.uint8)
    bufs = [torch.empty_like(send8) for _ in range(world)] if rank == dst else None
    dist.gather(send8, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r].view(torch.int16)[: sizes[r]] for r in range(world)], dim=0)


def compute_sharded(compute_fn, left, right, n_pairs, shape_hw, src=0, device=None, group=None):
    """scatter -> per-rank compute_fn(left_block, right_block) -> gather. Rank `src` passes the full batch, the
    others pass None. Returns the (N,H,W) int16 result on rank `src`."""
    l, r = scatter_pairs(left, right, n_pairs, shape_hw, src=src, device=device, group=group)
    if l.shape[0] > 0:
        d = compute_fn(l, r)
    else:
        d = torch.empty((0,) + tuple(shape_hw), dtype=torch.int16, device=l.device)
    return gather_disparities(d, n_pairs, dst=src, group=group)
