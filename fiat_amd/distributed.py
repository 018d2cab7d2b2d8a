"""Multi-GPU sharding of a batch of tabulation requests.

Requests are independent units: rank g of N owns the contiguous
block [g*ceil(n/N), ...) of the batch, tabulates it with no data-path collective,
and -- only when the consumer needs every table on every GPU -- the blocks are
reassembled with one all-gather (RCCL over xGMI on MI355X; backend "nccl" in
torch.distributed IS RCCL on ROCm; "gloo" is used by the CPU tests of this
plumbing).  One process per GPU.
"""
import torch
import torch.distributed as dist


def shard_bounds(nreq, rank, world):
    """Contiguous block of requests owned by ``rank``: equal blocks of
    ceil(nreq/world), the last ones possibly short or empty."""
    per = -(-nreq // world)
    lo = min(nreq, rank * per)
    hi = min(nreq, lo + per)
    return lo, hi


def tabulate_sharded(tabulate_fn, pts, verts=None, rank=None, world=None):
    """Run ``tabulate_fn(pts_block, verts_block)`` on this rank's block.
    Returns (local_tables, (lo, hi))."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(pts.shape[0], rank, world)
    local = tabulate_fn(pts[lo:hi], None if verts is None else verts[lo:hi])
    return local, (lo, hi)


def allgather_tables(local, nreq, group=None):
    """Reassemble the full (nreq, ...) table tensor on every rank from the
    per-rank blocks produced by ``tabulate_sharded``.  Equal-size blocks go
    through a single all_gather_into_tensor (one large collective: xGMI is
    point-to-point, few large transfers beat many small ones); a ragged tail is
    padded to the block size and trimmed afterwards."""
    world = dist.get_world_size(group)
    per = -(-nreq // world)
    tail_shape = tuple(local.shape[1:])
    if local.shape[0] != per:
        pad = torch.zeros((per,) + tail_shape, dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    full = torch.empty((world * per,) + tail_shape, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:nreq]
