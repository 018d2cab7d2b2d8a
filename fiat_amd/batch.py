"""A list of independent tabulation requests -> batched launches.

The unit of data parallelism of this library is the request (element, derivative order, evaluation points
[, cell vertices]) -- what one ``element.tabulate(order, points)`` call of the reference computes
(FIAT/finite_element.py:181-197).  Callers such as FInAT issue them one at a time per (element, point set)
(finat/fiat_elements.py:69); ``tabulate_requests`` takes the whole list, groups it by what a kernel launch needs to have in
common -- the element handle, the order, the number of points, whether cells are given -- stacks each
group's points into one device tensor, launches the groups round-robin on a few HIP streams (short groups then fill each
other's tails) and hands back, per request and in the caller's order, a VIEW of its group's output tensor.  With
``rank`` / ``world`` every group is cut into contiguous per-rank blocks first (fiat_amd/distributed.py): a rank tabulates its
blocks only and returns ``None`` for the requests it does not own."""
from collections import OrderedDict, namedtuple

import numpy
import torch

from . import runtime
from .distributed import shard_bounds

Request = namedtuple("Request", ["element", "order", "points", "verts"], defaults=[None])


def _group_key(req, npts):
    return (id(req.element), int(req.order), int(npts), req.verts is not None)


def tabulate_requests(requests, nstreams=4, rank=0, world=1, pushforward=False):
    """[Request] -> [device tensor (ntab, ndof, *value_shape, npts) | None], tables in mis() order."""
    ctx = runtime.Context.get()
    groups = OrderedDict()
    for index, req in enumerate(requests):
        if not isinstance(req, Request):
            req = Request(*req)
        pts = req.points
        npts = pts.shape[0]
        groups.setdefault(_group_key(req, npts), []).append((index, req))
    results = [None] * len(requests)
    main = torch.cuda.current_stream(ctx.device)
    pool = [torch.cuda.Stream(device=ctx.device) for _ in range(max(1, min(nstreams, len(groups))))]
    for s in pool:
        s.wait_stream(main)
    for g, members in enumerate(groups.values()):
        lo, hi = shard_bounds(len(members), rank, world)
        mine = members[lo:hi]
        if not mine:
            continue
        element, order = mine[0][1].element, mine[0][1].order

        def stacked(field):
            first = getattr(mine[0][1], field)
            if isinstance(first, torch.Tensor):
                return torch.stack([getattr(r, field).to(ctx.device, torch.float64) for _, r in mine])
            return numpy.stack([numpy.asarray(getattr(r, field), dtype=float) for _, r in mine])

        stream = pool[g % len(pool)]
        with torch.cuda.stream(stream):
            pts = runtime._as_device(stacked("points"), ctx)
            verts = runtime._as_device(stacked("verts"), ctx) if mine[0][1].verts is not None else None
            kwargs = {"verts": verts, "pushforward": pushforward} if verts is not None else {}
            out = element.tabulate_batch(order, pts, stream=stream, **kwargs)
        out.record_stream(main)     # allocated on the side stream, consumed on the caller's
        for row, (index, _) in enumerate(mine):
            results[index] = out[row]
    for s in pool:
        main.wait_stream(s)
    return results
