"""world_size-2 gloo test of the request sharding + all-gather plumbing (CPU).
The per-rank compute is the oracle here (tests may use it as the stand-in); on the
GPU box the same plumbing carries the HIP path's device tensors over RCCL."""
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


def _worker(rank, world, port, nreq, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fiat_amd_dist_shim import distributed as D
    rng = np.random.default_rng(0)
    e = rng.exponential(size=(nreq, 4, 3))
    pts = torch.as_tensor((e / e.sum(-1, keepdims=True))[..., 1:].copy())
    local, (lo, hi) = D.tabulate_sharded(_oracle_tabulate, pts)
    assert local.shape[0] == hi - lo
    full = D.allgather_tables(local, nreq)
    ref = _oracle_tabulate(pts, None)
    ok = bool(torch.equal(full, ref))
    q.put((rank, lo, hi, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nreq", [8, 7, 1])
def test_shard_and_allgather_world2(nreq, tmp_path):
    # import fiat_amd.distributed without importing the package __init__ (which
    # needs the built HIP library; here only the plumbing is under test)
    shim = tmp_path / "fiat_amd_dist_shim"
    shim.mkdir()
    (shim / "__init__.py").write_text("")
    src = open(os.path.join(ROOT, "fiat_amd", "distributed.py")).read()
    (shim / "distributed.py").write_text(src)
    sys.path.insert(0, str(tmp_path))
    os.environ["PYTHONPATH"] = str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", "")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nreq, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[3] for r in res] == [True, True]
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == nreq


def test_shard_bounds():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "fiat_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for n in (0, 1, 7, 8, 100000, 1000003):
        for w in (1, 2, 4, 8):
            b = [D.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= -(-n // w)
