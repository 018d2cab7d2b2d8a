#!/usr/bin/env python3
"""Expansion degrees 7 and 8: the stacked-matrix instances of round 3 against the generic kernel (measurement tooling)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd, bench
from fiat_amd import runtime

for fam, sd, deg in (("Lagrange", 3, 7), ("DiscontinuousLagrange", 3, 7), ("Lagrange", 2, 7), ("Lagrange", 2, 8)):
    cell = fiat_amd.ufc_simplex(sd)
    el = getattr(fiat_amd, fam)(cell, deg)
    ps = el.device_polyset()
    npts = len(fiat_amd.create_quadrature(cell, 2 * deg).get_points())
    for order in (0, 1, 2):
        shape1 = ps.out_shape(order, 1, npts)
        per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
        nreq = int(max(1, 0.8e9 // per_req))
        pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
        out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
        res = []
        for pol in ((), ("no_stacked",)):
            runtime.Context.get().set_policy(*pol)
            t = statistics.median(ps.time_tabulate_batch(order, pts, None, out, 5) for _ in range(3))
            res.append((ps.kernel_name(order, nreq, npts).split("::")[-1], t * 1e3, per_req * nreq / t / 1e6 / 80))
        runtime.Context.get().set_policy()
        print(f"{fam:22s} sd{sd} k{deg} order {order} npts {npts:3d}: " + "  |  ".join(f"{k}: {us:8.1f} us {fr:5.1f} % HBM" for k, us, fr in res), flush=True)
