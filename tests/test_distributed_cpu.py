"""world_size-2 gloo tests of the request sharding and of the three gather patterns of fiat_amd/distributed.py
(one-shot, chunked + overlapped, staging ring) on CPU.  The per-rank compute is the oracle here (tests may use it
as the stand-in); on a GPU node the same code carries the HIP path's device tensors over RCCL through the C ABI
(fx_allgather_tables)."""
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
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_tabulate(pts, verts):
    from oracle import fiat_oracle as fo
    co, _, _ = fo.lagrange_coeffs(fo.UFC_SIMPLEX[2], 2)
    out = []
    for p in pts.numpy():
        tab = fo.element_tabulate(fo.UFC_SIMPLEX[2], 2, co, 1, p, 1, "bubble")
        out.append(np.stack([tab[a] for a in fo.jet_indices(2, 1)]))
    if not out:
        return torch.zeros((0, 3, 6, pts.shape[1]), dtype=torch.float64)
    return torch.as_tensor(np.stack(out))


def _worker(rank, world, port, nreq, chunk, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fiat_amd import distributed as D
    rng = np.random.default_rng(0)
    e = rng.exponential(size=(nreq, 4, 3))
    pts = torch.as_tensor((e / e.sum(-1, keepdims=True))[..., 1:].copy())
    ref = _oracle_tabulate(pts, None)
    gather = D.TableGather()
    assert gather.impl == "torch" and gather.world == world
    # 1. one exchange of (padded) equal blocks
    local, (lo, hi) = D.tabulate_sharded(_oracle_tabulate, pts)
    assert local.shape[0] == hi - lo
    ok = [bool(torch.equal(gather.all_gather(local, nreq), ref))]
    ok.append(bool(torch.equal(D.allgather_tables(local, nreq), ref)))
    # 2. chunked gather, a chunk exchanged as soon as it is produced
    per = -(-nreq // world)
    full = torch.full((world * per,) + tuple(ref.shape[1:]), float("nan"), dtype=torch.float64)
    calls = []

    def produce(c0, c1, rows):
        calls.append((c0, c1))
        rows.copy_(_oracle_tabulate(pts[lo + c0:lo + c1], None))

    gather.tabulate_allgather(produce, hi - lo, per, chunk, full)
    ok.append(bool(torch.equal(full[:nreq], ref)))
    ok.append(calls == [(c, min(c + chunk, hi - lo)) for c in range(0, hi - lo, chunk)])
    # 3. staging ring for outputs too large to replicate: the consumer sees every chunk of every rank once
    padded = torch.zeros((per,) + tuple(ref.shape[1:]), dtype=torch.float64)
    padded[: hi - lo] = local
    seen = torch.full_like(full, float("nan")).view(world, per, *ref.shape[1:])
    nchunks = 0
    for c0, c1, staged in gather.iter_gathered_chunks(padded, chunk, ring=2):
        assert staged.shape[:2] == (world, c1 - c0)
        seen[:, c0:c1] = staged
        nchunks += 1
    ok.append(nchunks == -(-per // chunk))
    ok.append(bool(torch.equal(seen.reshape(world * per, *ref.shape[1:])[:nreq], ref)))
    gather.close()
    q.put((rank, lo, hi, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nreq,chunk", [(8, 2), (7, 3), (1, 1), (13, 4)])
def test_shard_and_gather_world2(nreq, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nreq, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(all(r[3]) for r in res), res
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == nreq


def test_shard_bounds():
    from fiat_amd import distributed as D
    for n in (0, 1, 7, 8, 100000, 1000003):
        for w in (1, 2, 4, 8):
            b = [D.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= -(-n // w)


# ---- bench.py --allgather-verify: the data check of the exchange, rehearsed with two gloo ranks on CPU ---------------
class _FakeWorkload:
    """What verify_gather needs of a bench workload: per-request inputs and produce_rows (a deterministic function of
    the inputs standing in for the tabulation)."""

    def __init__(self, rank, nreq):
        g = torch.Generator().manual_seed(100 + rank)          # ranks draw DIFFERENT inputs, as in the bench
        self.parts = [{"pts": torch.rand((nreq, 3, 2), dtype=torch.float64, generator=g)}]

    def produce_rows(self, lo, hi, rows):
        p = self.parts[0]["pts"][lo:hi]
        rows.copy_(torch.einsum("rpd,rqe->rpqde", p, p).reshape(hi - lo, -1)[:, :12].reshape(hi - lo, 3, 4))


def _verify_worker(rank, world, port, sabotage, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from fiat_amd import distributed as D
    gather = D.TableGather()
    if sabotage:                       # an exchange that lands foreign blocks one row off: the check must see it
        good = gather._exchange

        def bad(send, full, per, offset, stream=None):
            good(send, full, per, offset, stream)
            view = full.view(world, per, *full.shape[1:])
            other = 1 - rank
            view[other, offset:offset + send.shape[0]] = torch.roll(view[other, offset:offset + send.shape[0]], 1, 0)
        gather._exchange = bad
    nreq = 37
    wl = _FakeWorkload(rank, nreq)
    before = wl.parts[0]["pts"].clone()
    out = torch.empty((nreq, 3, 4), dtype=torch.float64)
    res = bench.verify_gather(wl, gather, world, rank, nreq, out, (3, 4))
    restored = bool(torch.equal(before, wl.parts[0]["pts"]))
    gather.close()
    q.put((rank, res, restored))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sabotage", [False, True])
def test_allgather_verify_world2(sabotage):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, sabotage, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r, restored in res:
        assert restored                                        # the workload's own inputs are put back
        assert r["requests"] == 37 and r["ok"] == (not sabotage), r
        if sabotage:
            assert not r["one_shot_blocks_equal"][1 - rank] and r["one_shot_blocks_equal"][rank]
            # ... and says where, for every rank's view, in the one dict rank 0 prints: the foreign block, from its first
            # request on (the sabotage rolls whole requests by one), none of it unwritten, and the shift it amounts to
            mm = r["mismatches"]
            assert {(m["seen_by_rank"], m["peer_block"]) for m in mm} == {(0, 1), (1, 0)}, mm
            one = [m for m in mm if m["leg"] == "one_shot" and m["seen_by_rank"] == rank][0]
            assert one["peer_block"] == 1 - rank and one["first_request"] == 0 and one["never_written"] == 0
            assert one["differing"] > 0 and one["doubles_per_block"] == 37 * 12
            assert any(m["leg"] == "chunked" and m.get("chunk_requests") == 37 // 5 + 1 for m in mm)
        else:
            assert "mismatches" not in r
            assert all(r["one_shot_blocks_equal"]) and all(r["chunked_blocks_equal"])
