#!/usr/bin/env python3
"""Per-request-cell launches of chosen shapes, one line each: registry instance, launch time, fraction of the HBM peak
(measurement tooling).  python tools/instance_ab.py "Lagrange,3,4,44,1;RaviartThomas,3,3,23,2" [--policy no_small] [--pushforward]
(family, sd, degree, points, derivative order).  A/B between libraries: run it under FIAT_AMD_LIB=... alternately
(tools/tool_ab.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fiat_amd, bench
from fiat_amd import runtime
if "--policy" in sys.argv:
    runtime.Context.get().set_policy(*sys.argv[sys.argv.index("--policy") + 1].split(","))
cells = "--own-cell" not in sys.argv
for spec in sys.argv[1].split(";"):
    fam, sd, deg, npts, order = spec.split(",")
    sd, deg, npts, order = int(sd), int(deg), int(npts), int(order)
    cell = fiat_amd.ufc_simplex(sd)
    el = getattr(fiat_amd, fam)(cell, deg)
    ps = el.device_polyset()
    shape1 = ps.out_shape(order, 1, npts)
    per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
    nreq = int(min(2_000_000, float(os.environ.get("CAP_GB", "0.8")) * 1e9 // per_req))
    pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
    verts = None
    if cells:
        rng = np.random.default_rng(3)
        A = torch.as_tensor(np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))).cuda()
        b = torch.as_tensor(rng.standard_normal((nreq, 1, sd))).cuda()
        ref = torch.as_tensor(np.array(cell.get_vertices(), dtype=float)).cuda()
        verts = (torch.einsum("vd,red->rve", ref, A) + b).contiguous()
        pts = (torch.einsum("rpd,red->rpe", pts, A) + b).contiguous()
    out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
    name = ps.kernel_name(order, nreq, npts, has_verts=cells, instance=True)
    if "--pushforward" in sys.argv and cells and el.mapping()[0] != "affine":   # the element's Piola map with the tabulation
        m = el.mapping()[0]
        name = ps.kernel_name(order, nreq, npts, has_verts=True, instance=True, mapping=m)

        def timed(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ps.tabulate_batch(order, pts, verts=verts, out=out, mapping=m)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        timed(5)
        t = min(timed(10) for _ in range(5))
    else:
        for _ in range(3):
            ps.time_tabulate_batch(order, pts, verts, out, 5)
        t = min(ps.time_tabulate_batch(order, pts, verts, out, 10) for _ in range(5))
    print(f"{spec:34s} {t * 1e3:8.1f} us  {nreq * per_req / t / 1e6 / 8000 * 100:5.1f} % HBM  {name.replace('fxk::tabulate_simplex_', '')}", flush=True)
