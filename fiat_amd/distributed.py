"""Multi-GPU sharding of a batch of tabulation requests.

Requests are independent units: rank g of N (one process per GPU) owns the contiguous block
``shard_bounds(nreq, g, N)`` and tabulates it with NO data-path collective.  Only when the consumer needs every
table on every GPU are the blocks exchanged, and then

* through the C ABI (``fx_allgather_tables``: RCCL over xGMI, called from C++; "direct" = one grouped send/recv
  with every peer, all 7 links of a GPU at once -- the node is fully connected and a ring is bound by one link),
* chunk by chunk on a second HIP stream while the next chunk is being tabulated (``tabulate_allgather``), or
* for outputs too large to replicate (C4 at 122 points: 823 GB) through a ring of staging buffers that the consumer
  drains chunk by chunk (``iter_gathered_chunks``).

``torch.distributed`` is plumbing: rendezvous, the exchange of the 128-byte communicator id, barriers.  With the
"gloo" backend (CPU tests of this module, rehearsals on a one-GPU box) the same calls run through
``torch.distributed`` collectives on host memory.
"""
import ctypes

import torch
import torch.distributed as dist


def shard_bounds(nreq, rank, world):
    """Contiguous block of requests owned by ``rank``: equal blocks of ceil(nreq / world), the last ones possibly
    short or empty."""
    per = -(-nreq // world)
    lo = min(nreq, rank * per)
    hi = min(nreq, lo + per)
    return lo, hi


def tabulate_sharded(tabulate_fn, pts, verts=None, rank=None, world=None):
    """Run ``tabulate_fn(pts_block, verts_block)`` on this rank's block.  Returns (local_tables, (lo, hi))."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(pts.shape[0], rank, world)
    local = tabulate_fn(pts[lo:hi], None if verts is None else verts[lo:hi])
    return local, (lo, hi)


class TableGather:
    """Exchange of per-rank table blocks.  ``impl``: "rccl" (C ABI), "torch" (torch.distributed collectives) or
    "auto" = rccl when the process group runs on RCCL and every rank could load it."""

    def __init__(self, group=None, impl="auto", algo="direct", device=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.algo = algo
        self.comm = None
        self._comm_stream = None
        backend = dist.get_backend(group)
        want = impl == "rccl" or (impl == "auto" and backend == "nccl")
        if want:
            from . import _lib, runtime
            self._lib, self._runtime = _lib, runtime
            self.ctx = runtime.Context.get(device)
            # communicator creation is collective: agree first that every rank can load RCCL
            ok = torch.tensor([1 if _lib.lib.fx_comm_available() == 0 else 0], dtype=torch.int32,
                              device=self.ctx.device if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 1:
                ident = [None]
                if self.rank == 0:
                    buf = (ctypes.c_ubyte * 128)()
                    _lib.check(_lib.lib.fx_comm_unique_id(buf))
                    ident[0] = bytes(buf)
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group else 0, group=group)
                raw = (ctypes.c_ubyte * 128).from_buffer_copy(ident[0])
                handle = ctypes.c_void_p()
                _lib.check(_lib.lib.fx_comm_create(self.ctx.handle, self.world, self.rank, raw, ctypes.byref(handle)))
                self.comm = handle
            elif impl == "rccl":
                raise RuntimeError("RCCL could not be loaded on every rank: " + _lib.lib.fx_last_error().decode())
        self.impl = "rccl" if self.comm is not None else "torch"

    def close(self):
        """Destroys the communicator.  fx_comm_destroy first waits for the streams exchanges were enqueued on: the
        gathered tables are complete when this returns."""
        if self.comm is not None:
            comm, self.comm = self.comm, None
            self._lib.check(self._lib.lib.fx_comm_destroy(comm))

    # -- one block per rank into recv[p * stride + offset] ------------------------------------------------
    def _exchange(self, send, full, per, offset, stream=None):
        """``send`` (m, *tail): this rank's rows [offset, offset + m) of its block; ``full`` (world * per, *tail)."""
        m = send.shape[0]
        row = int(send[0].numel()) if m else 0
        if m == 0:
            return
        if self.impl == "rccl":
            if not (send.is_contiguous() and full.is_contiguous()):
                raise ValueError("table blocks must be contiguous")
            ring = self.algo == "ring" and m == per and offset == 0
            if full.shape[0] < self.world * per or (m and int(full[0].numel()) != row):
                raise ValueError(f"receive buffer of {tuple(full.shape)} cannot hold {self.world} blocks of {per} rows of {row}")
            self._lib.check(self._lib.lib.fx_allgather_tables(
                self.comm, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(full.data_ptr()), m * row, per * row,
                offset * row, int(full.numel()), 0 if ring else 1, self._runtime._stream_ptr(stream)))
            return
        # torch.distributed collectives (gloo: through host memory)
        host = dist.get_backend(self.group) != "nccl"
        src = send.contiguous()
        staged = torch.empty((self.world * m,) + tuple(src.shape[1:]), dtype=src.dtype, device="cpu" if host else src.device)
        dist.all_gather_into_tensor(staged, src.cpu() if host else src, group=self.group)
        view = full.view(self.world, per, *full.shape[1:])
        view[:, offset:offset + m] = staged.view(self.world, m, *src.shape[1:]).to(full.device)

    def _padded(self, local, per):
        if local.shape[0] == per:
            return local
        pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        return pad

    def all_gather(self, local, nreq, out=None, stream=None):
        """The full (nreq, ...) tensor on every rank from the blocks ``tabulate_sharded`` produced (one exchange of
        equal blocks; a ragged tail is padded to the block size and trimmed afterwards)."""
        per = -(-nreq // self.world)
        tail = tuple(local.shape[1:])
        full = out if out is not None else torch.empty((self.world * per,) + tail, dtype=local.dtype, device=local.device)
        if tuple(full.shape) != (self.world * per,) + tail:
            raise ValueError(f"out must have shape {(self.world * per,) + tail}")
        self._exchange(self._padded(local, per), full, per, 0, stream)
        return full[:nreq]

    def tabulate_allgather(self, produce, nlocal, per, chunk, full):
        """Chunked gather overlapped with compute.  ``produce(lo, hi, rows)`` enqueues, on the current stream, the
        tabulation of this rank's requests [lo, hi) into ``rows`` (a view of this rank's block of ``full``); as soon
        as a chunk is produced its exchange starts on a second stream while the next chunk is computed.  ``full``:
        (world * per, *tail), complete on return (in stream order)."""
        mine = full.view(self.world, per, *full.shape[1:])[self.rank]
        gpu = full.is_cuda
        if gpu:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=full.device)
            main = torch.cuda.current_stream(full.device)
        for c0 in range(0, per, chunk):
            c1 = min(per, c0 + chunk)
            if c0 < nlocal:
                produce(c0, min(c1, nlocal), mine[c0:min(c1, nlocal)])
            if gpu:
                done = torch.cuda.Event()
                done.record(main)
                self._comm_stream.wait_event(done)
                with torch.cuda.stream(self._comm_stream):
                    self._exchange(mine[c0:c1], full, per, c0, self._comm_stream)
            else:
                self._exchange(mine[c0:c1], full, per, c0)
        if gpu:
            main.wait_stream(self._comm_stream)
        return full

    def iter_gathered_chunks(self, local, chunk, ring=2):
        """For outputs that cannot be replicated whole: yields (lo, hi, staged) with ``staged[p]`` = rows [lo, hi) of
        rank p's block, gathered into a ring of ``ring`` staging buffers of ``chunk`` rows per rank.  A staging buffer
        is reused ``ring`` chunks later: the consumer must have enqueued its reads of a chunk (on the current stream)
        before it asks for the next one.  ``local`` must have the same number of rows on every rank (pad the last
        block): the exchange is collective."""
        per = local.shape[0]
        tail = tuple(local.shape[1:])
        bufs = [torch.empty((self.world * chunk,) + tail, dtype=local.dtype, device=local.device) for _ in range(ring)]
        gpu = local.is_cuda
        if gpu and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=local.device)
        starts = list(range(0, per, chunk))
        ready, drained = {}, {}

        def issue(i):
            c0 = starts[i]
            piece = self._padded(local[c0:c0 + chunk], chunk)
            if gpu:
                self._comm_stream.wait_stream(torch.cuda.current_stream(local.device))
                if i - ring in drained:
                    self._comm_stream.wait_event(drained[i - ring])
                with torch.cuda.stream(self._comm_stream):
                    self._exchange(piece, bufs[i % ring], chunk, 0, self._comm_stream)
                    ready[i] = torch.cuda.Event()
                    ready[i].record(self._comm_stream)
            else:
                self._exchange(piece, bufs[i % ring], chunk, 0)

        for i in range(min(ring - 1, len(starts))):
            issue(i)
        for i, c0 in enumerate(starts):
            if i + ring - 1 < len(starts):
                issue(i + ring - 1)
            if gpu:
                torch.cuda.current_stream(local.device).wait_event(ready.pop(i))
            m = min(chunk, per - c0)
            yield c0, c0 + m, bufs[i % ring].view(self.world, chunk, *tail)[:, :m]
            if gpu:
                drained[i] = torch.cuda.Event()
                drained[i].record(torch.cuda.current_stream(local.device))


def allgather_tables(local, nreq, group=None, impl="auto"):
    """One-shot convenience: the full (nreq, ...) table tensor on every rank."""
    gather = TableGather(group, impl=impl)
    try:
        return gather.all_gather(local, nreq)
    finally:
        gather.close()
