#!/usr/bin/env python3
"""Time one raw polynomial set (random coefficients: no element construction, so ablation builds work) over request shapes
(measurement tooling).  FIAT_AMD_LIB=.../libfiat_amd_dbgN.so python tools/stacked_ablation.py"""
import os, sys, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fiat_amd import runtime
rng = np.random.default_rng(0)
for sd, n, npts in ((3, 6, 122), (3, 6, 23), (3, 5, 74), (3, 5, 23), (3, 4, 44), (2, 6, 33)):
    nexp = math.comb(n + sd, sd)
    ps = runtime.SimplexPolySet(sd, n, coeffs=rng.standard_normal((nexp, nexp)))
    for order in (0, 1, 2):
        shape1 = ps.out_shape(order, 1, npts)
        per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
        nreq = int(0.8e9 // per_req)
        pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
        out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
        t = statistics.median(ps.time_tabulate_batch(order, pts, None, out, 5) for _ in range(3))
        print(f"sd{sd} n{n} order {order} npts {npts:3d}: {t*1e3:8.1f} us  {per_req*nreq/t/1e6/80:5.1f} % HBM  {ps.kernel_name(order, nreq, npts)}", flush=True)
        del pts, out
