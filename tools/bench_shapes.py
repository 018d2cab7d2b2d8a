#!/usr/bin/env python3
"""HBM fraction of tabulate_batch for a range of common element shapes (measurement tooling)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd, bench

CASES = [("Lagrange", 2, 1, 3), ("Lagrange", 2, 2, 6), ("Lagrange", 2, 3, 12), ("Lagrange", 3, 1, 4), ("Lagrange", 3, 2, 11),
         ("Lagrange", 3, 2, 14), ("Lagrange", 3, 3, 23), ("Lagrange", 3, 3, 14), ("Lagrange", 2, 4, 12), ("Lagrange", 2, 5, 16), ("Lagrange", 3, 4, 23), ("DiscontinuousLagrange", 3, 1, 4),
         ("Nedelec", 3, 1, 4), ("RaviartThomas", 3, 1, 4), ("Nedelec", 2, 1, 3)]
# (family, sd, degree, points, order): larger shapes, --no-stacked / --no-fixed / --no-small for the A/B partners
BIG = [("Lagrange", 3, 4, 23, 2), ("Lagrange", 3, 5, 23, 1), ("Lagrange", 3, 5, 23, 2), ("DiscontinuousLagrange", 3, 5, 30, 1),
       ("DiscontinuousLagrange", 3, 6, 23, 1), ("DiscontinuousLagrange", 3, 6, 40, 2), ("Lagrange", 3, 6, 23, 2),
       ("Nedelec", 3, 4, 23, 1), ("Nedelec", 3, 3, 23, 1), ("BrezziDouglasMarini", 3, 3, 23, 1)]
MID = [("Lagrange", 2, 5, 16, 1), ("Lagrange", 2, 5, 23, 2), ("Lagrange", 2, 6, 23, 1), ("Lagrange", 2, 6, 23, 2),
       ("DiscontinuousLagrange", 2, 6, 30, 2), ("Nedelec", 2, 5, 23, 1), ("Lagrange", 3, 4, 30, 1), ("Lagrange", 3, 5, 23, 0),
       ("Lagrange", 3, 3, 23, 2), ("RaviartThomas", 3, 3, 23, 0), ("Lagrange", 3, 6, 23, 0)]
MANY = [("DiscontinuousLagrange", 3, 6, 122, 2), ("Lagrange", 3, 5, 97, 1), ("Lagrange", 3, 4, 122, 2), ("Lagrange", 3, 3, 70, 1),
        ("Lagrange", 2, 6, 73, 2), ("Nedelec", 3, 3, 100, 1)]
if "--many-points" in sys.argv:
    CASES = MANY
if "--big" in sys.argv:
    CASES = BIG
if "--mid" in sys.argv:
    CASES = MID
from fiat_amd import runtime
runtime.Context.get().set_policy(*[f[2:].replace("-", "_") for f in sys.argv if f in ("--no-stacked", "--no-fixed", "--no-small", "--no-coop")])
for case in CASES:
    fam, sd, deg, npts = case[:4]
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
    ps = el.device_polyset()
    order = case[4] if len(case) > 4 else 1
    shape1 = ps.out_shape(order, 1, npts)
    per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
    nreq = int(min(4_000_000, 1.2e9 // per_req))
    pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
    verts = None
    if "--verts" in sys.argv:  # per-request cells: affine images of the reference cell, points mapped along
        rng = np.random.default_rng(3)
        A = torch.as_tensor(np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))).cuda()
        b = torch.as_tensor(rng.standard_normal((nreq, 1, sd))).cuda()
        ref = torch.as_tensor(np.array(fiat_amd.ufc_simplex(sd).get_vertices(), dtype=float)).cuda()
        verts = (torch.einsum("vd,red->rve", ref, A) + b).contiguous()
        pts = (torch.einsum("rpd,red->rpe", pts, A) + b).contiguous()
    out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
    t = statistics.median(ps.time_tabulate_batch(order, pts, verts, out, 10) for _ in range(3))
    print(f"{fam:22s} sd{sd} P{deg} npts {npts:3d}: {nreq:8d} requests, {t*1e3:8.1f} us, {nreq/t/1e3:8.1f} M req/s, "
          f"{per_req*nreq/t/1e6:6.0f} GB/s ({per_req*nreq/t/1e6/80:.0f} %)  {ps.kernel_name(order, nreq, npts, has_verts=verts is not None)}")
    del pts, out
