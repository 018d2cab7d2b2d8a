#!/usr/bin/env python3
"""Lane-local kernel against the stacked-matrix kernel on the shapes both can serve (measurement tooling):
python tools/small_vs_stacked.py [--verts]  -> % of the HBM peak under the default policy and under no_stacked."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd, bench
from fiat_amd import runtime
ctx = runtime.Context.get()
CASES = [(3, 2, (4, 8, 11, 14, 16, 24, 31)), (2, 3, (6, 9, 12, 16, 25)), (2, 4, (6, 12, 16, 25, 33))]
for sd, deg, nps in CASES:
    cell = fiat_amd.ufc_simplex(sd)
    for fam in ("Lagrange", "RaviartThomas"):
        el = getattr(fiat_amd, fam)(cell, deg)
        ps = el.device_polyset()
        for npts in nps:
            for order in (0, 1, 2):
                shape1 = ps.out_shape(order, 1, npts)
                per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
                nreq = int(min(2_000_000, 0.8e9 // per_req))
                pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
                verts = None
                if "--verts" in sys.argv:
                    rng = np.random.default_rng(3)
                    A = torch.as_tensor(np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))).cuda()
                    b = torch.as_tensor(rng.standard_normal((nreq, 1, sd))).cuda()
                    ref = torch.as_tensor(np.array(cell.get_vertices(), dtype=float)).cuda()
                    verts = (torch.einsum("vd,red->rve", ref, A) + b).contiguous()
                    pts = (torch.einsum("rpd,red->rpe", pts, A) + b).contiguous()
                out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
                res = []
                for pol in ((), ("no_stacked",), ("no_stacked", "no_small")):
                    ctx.set_policy(*pol)
                    kern = ps.kernel_name(order, nreq, npts, has_verts=verts is not None).replace("fxk::tabulate_simplex_", "")
                    t = statistics.median(ps.time_tabulate_batch(order, pts, verts, out, 5) for _ in range(3))
                    res.append(f"{kern:8s} {per_req * nreq / t / 1e6 / 80:5.1f} %")
                ctx.set_policy()
                print(f"{fam:14s} sd{sd} k{deg} order {order} npts {npts:3d} rows {int(np.prod(shape1[2:-1])):3d}:  " + "   ".join(res), flush=True)
                del pts, out
