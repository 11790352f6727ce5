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


def _p2p(ops, group):
    """One batch of point-to-point transfers (an ncclGroupStart/End of ncclSend/ncclRecv on RCCL). Returns the requests."""
    return dist.batch_isend_irecv(ops) if ops else []


def _wait(reqs):
    for r in reqs:
        r.wait()


def scatter_pairs(left, right, n_pairs, shape_hw, src=0, device=None, group=None):
    """Rank `src` holds left/right (N,H,W) uint8; every rank returns its (n_r,H,W) block. Point-to-point sends of the
    blocks themselves (views of the batch: no padded copies on the root, ragged blocks travel at their own size)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes = shard_sizes(n_pairs, world)
    h, w = shape_hw
    if device is None:
        device = left.device if left is not None else torch.device("cpu")
    if rank == src:
        ops = []
        for r in range(world):
            lo, hi = shard_bounds(n_pairs, r, world)
            if r != src and hi > lo:
                # the transport moves tensors of `device` (nccl: this rank's GPU): a batch that lives elsewhere is staged
                # block by block; .to() is a no-op for a batch that is already there
                ops += [dist.P2POp(dist.isend, left[lo:hi].to(device), r, group), dist.P2POp(dist.isend, right[lo:hi].to(device), r, group)]
        reqs = _p2p(ops, group)
        lo, hi = shard_bounds(n_pairs, src, world)
        mine = left[lo:hi].to(device).contiguous(), right[lo:hi].to(device).contiguous()
        _wait(reqs)
        return mine
    l = torch.empty((sizes[rank], h, w), dtype=torch.uint8, device=device)
    r_ = torch.empty((sizes[rank], h, w), dtype=torch.uint8, device=device)
    if sizes[rank] > 0:
        _wait(_p2p([dist.P2POp(dist.irecv, l, src, group), dist.P2POp(dist.irecv, r_, src, group)], group))
    return l, r_


def gather_disparities(local_disp, n_pairs, dst=0, group=None):
    """Inverse of scatter_pairs for the int16 maps. Returns (N,H,W) on rank `dst`, None elsewhere. The maps travel as
    raw bytes (gloo has no int16 transport) straight into their slice of the result."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    h, w = local_disp.shape[-2:]
    if rank != dst:
        if local_disp.shape[0] > 0:
            _wait(_p2p([dist.P2POp(dist.isend, local_disp.contiguous().view(torch.uint8), dst, group)], group))
        return None
    out = torch.empty((n_pairs, h, w), dtype=torch.int16, device=local_disp.device)
    ops = []
    for r in range(world):
        lo, hi = shard_bounds(n_pairs, r, world)
        if r != dst and hi > lo:
            ops.append(dist.P2POp(dist.irecv, out[lo:hi].view(torch.uint8), r, group))
    reqs = _p2p(ops, group)
    lo, hi = shard_bounds(n_pairs, dst, world)
    out[lo:hi] = local_disp
    _wait(reqs)
    return out


def compute_sharded(compute_fn, left, right, n_pairs, shape_hw, src=0, device=None, group=None):
    """scatter -> per-rank compute_fn(left_block, right_block) -> gather. Rank `src` passes the full batch, the
    others pass None. Returns the (N,H,W) int16 result on rank `src`."""
    l, r = scatter_pairs(left, right, n_pairs, shape_hw, src=src, device=device, group=group)
    if l.shape[0] > 0:
        d = compute_fn(l, r)
    else:
        d = torch.empty((0,) + tuple(shape_hw), dtype=torch.int16, device=l.device)
    return gather_disparities(d, n_pairs, dst=src, group=group)


def compute_sharded_chunked(compute_fn, left, right, n_pairs, shape_hw, chunk=8, src=0, device=None, group=None, out=None):
    """Same result as compute_sharded, but every rank's block moves in chunks of `chunk` pairs and the transfers are
    double-buffered against the computation ("chunk (e.g. 8 pairs) and double-buffer"):

        step t:  inputs of chunk t travel root -> peers | chunk t-1 is computed | maps of chunk t-2 travel peers -> root

    One batch of point-to-point operations per step (ncclSend/ncclRecv inside one group on RCCL, so the root's seven
    xGMI links carry their peers' chunks concurrently); root and peers post the same sequence, which is what keeps the
    send/recv queues of a pair of ranks from blocking each other. The root sends views of the caller's batch and
    receives straight into the result: no padded or duplicated copies. A ragged last chunk travels at its own size.
    `compute_fn(l, r)` must have finished (its result usable) when it returns. Returns (N,H,W) int16 on `src`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    h, w = shape_hw
    if device is None:
        device = left.device if left is not None else torch.device("cpu")
    bounds = [shard_bounds(n_pairs, r, world) for r in range(world)]

    def chunk_span(r, t):            # pairs [a, b) of chunk t of rank r (empty when the block has fewer chunks)
        lo, hi = bounds[r]
        a = min(hi, lo + t * chunk)
        return a, min(hi, a + chunk)

    nsteps = max((hi - lo + chunk - 1) // chunk for lo, hi in bounds)
    lo, hi = bounds[rank]
    if rank == src:
        if out is None:
            out = torch.empty((n_pairs, h, w), dtype=torch.int16, device=device)
        pend = []
        for t in range(nsteps + 2):
            ops = []
            for r in range(world):
                if r == src:
                    continue
                a, b = chunk_span(r, t)
                if b > a:
                    ops += [dist.P2POp(dist.isend, left[a:b].to(device), r, group), dist.P2POp(dist.isend, right[a:b].to(device), r, group)]
                a, b = chunk_span(r, t - 2)
                if t >= 2 and b > a:
                    ops.append(dist.P2POp(dist.irecv, out[a:b].view(torch.uint8), r, group))
            reqs = _p2p(ops, group)
            a, b = chunk_span(src, t - 1)          # the root's own chunk of this step is computed under the transfers
            if t >= 1 and b > a:
                out[a:b] = compute_fn(left[a:b], right[a:b])
            pend.append(reqs)
            if len(pend) > 2:                      # keep two steps of transfers in flight
                _wait(pend.pop(0))
        for reqs in pend:
            _wait(reqs)
        return out
    cmax = min(chunk, max(hi - lo, 1))
    inbuf = [(torch.empty((cmax, h, w), dtype=torch.uint8, device=device), torch.empty((cmax, h, w), dtype=torch.uint8, device=device))
             for _ in range(2)]
    outbuf = [torch.empty((cmax, h, w), dtype=torch.int16, device=device) for _ in range(2)]
    step_req = [[], []]                            # requests of the last two steps
    for t in range(nsteps + 2):
        ops = []
        a, b = chunk_span(rank, t)
        if b > a:                                  # inputs of chunk t -> inbuf[t % 2] (chunk t-2 has been computed)
            l, r_ = inbuf[t % 2]
            ops += [dist.P2POp(dist.irecv, l[: b - a], src, group), dist.P2POp(dist.irecv, r_[: b - a], src, group)]
        a2, b2 = chunk_span(rank, t - 2)
        if t >= 2 and b2 > a2:                     # maps of chunk t-2 leave from outbuf[(t - 2) % 2]
            ops.append(dist.P2POp(dist.isend, outbuf[t % 2][: b2 - a2].view(torch.uint8), src, group))
        reqs = _p2p(ops, group)
        a1, b1 = chunk_span(rank, t - 1)
        if t >= 1 and b1 > a1:
            # step t-1 carried the inputs of chunk t-1 and the send that last used outbuf[(t - 1) % 2] (chunk t-3)
            _wait(step_req[(t - 1) % 2])
            step_req[(t - 1) % 2] = []
            l, r_ = inbuf[(t - 1) % 2]
            outbuf[(t - 1) % 2][: b1 - a1] = compute_fn(l[: b1 - a1], r_[: b1 - a1])
        _wait(step_req[t % 2])                     # (left over when a step had nothing to compute)
        step_req[t % 2] = reqs
    for reqs in step_req:
        _wait(reqs)
    return None
