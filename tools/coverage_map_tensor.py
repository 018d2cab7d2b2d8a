#!/usr/bin/env python3
"""HBM fraction of the fused tensor-product kernel over quadrilateral / hexahedron shapes (measurement tooling):
Q_k elements, k = 1..4, derivative orders 0..2, tensor grid of (k + 1)^d points per request and the same number of scattered points.  CAP_GB=8: larger batches."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd
I = fiat_amd.ufc_simplex(1)
rng = np.random.default_rng(5)
for dim in (2, 3):
    for k in (1, 2, 3, 4):
        P = fiat_amd.Lagrange(I, k)
        el = fiat_amd.TensorProductElement(P, P)
        if dim == 3:
            el = fiat_amd.TensorProductElement(el, P)
        q = k + 1
        npts, ndof = q ** dim, (k + 1) ** dim
        for order in (0, 1, 2):
            ntab = sum(1 for a in range(order + 1) for _ in range(1)) if dim == 1 else (
                (order + 1) * (order + 2) // 2 if dim == 2 else (order + 1) * (order + 2) * (order + 3) // 6)
            per = 8 * ntab * ndof * npts
            nreq = int(min(4_000_000, float(os.environ.get('CAP_GB', '0.8')) * 1e9 // per))
            out = torch.empty((nreq, ntab, ndof, npts), dtype=torch.float64, device="cuda")
            for mode in ("grid", "points"):
                if mode == "grid":
                    x = torch.as_tensor(np.sort(rng.uniform(0, 1, size=(nreq, dim, q)), axis=2)).cuda()
                    fn = lambda: el.tabulate_batch(order, x, out=out, grid=True)
                else:
                    x = torch.as_tensor(rng.uniform(0, 1, size=(nreq, npts, dim))).cuda()
                    fn = lambda: el.tabulate_batch(order, x, out=out)
                for _ in range(2): fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5): fn()
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 5)
                ms = statistics.median(ts)
                print(f"Q{k} dim {dim} order {order} {mode:6s}: npts {npts:3d} ndof {ndof:3d} {nreq:8d} req {ms*1e3:8.1f} us {per*nreq/ms/1e6/80:5.1f} % HBM", flush=True)
                del x
            del out
