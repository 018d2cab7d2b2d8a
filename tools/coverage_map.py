#!/usr/bin/env python3
"""Which kernel serves which realistic request shape, and how close to the HBM / fp64 rooflines it runs (measurement tooling):
families x cell x degree x derivative order, points = the default quadrature rule of degree 2 * degree (what a mass /
stiffness assembly asks for).  python tools/coverage_map.py [--verts [--pushforward]] [--order K] [--policy no_small,no_stacked] [--only "Lagrange sd3"]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd, bench

FAMS = [("Lagrange", range(1, 7)), ("DiscontinuousLagrange", range(0, 7)), ("Nedelec", range(1, 5)), ("RaviartThomas", range(1, 5)),
        ("BrezziDouglasMarini", range(1, 4)), ("NedelecSecondKind", range(1, 4))]
orders = [int(sys.argv[sys.argv.index("--order") + 1])] if "--order" in sys.argv else [0, 1, 2]
from fiat_amd import runtime
if "--policy" in sys.argv:
    runtime.Context.get().set_policy(*sys.argv[sys.argv.index("--policy") + 1].split(","))
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
rows = []
for sd in (2, 3):
    cell = fiat_amd.ufc_simplex(sd)
    for fam, degs in FAMS:
        for deg in degs:
            if sd == 3 and fam in ("Nedelec", "RaviartThomas") and deg > 3:
                continue
            if only and only not in f"{fam} sd{sd} k{deg}":
                continue
            el = getattr(fiat_amd, fam)(cell, deg)
            ps = el.device_polyset()
            npts = len(fiat_amd.create_quadrature(cell, max(1, 2 * deg)).get_points())
            for order in orders:
                shape1 = ps.out_shape(order, 1, npts)
                per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
                nreq = int(min(2_000_000, float(os.environ.get('CAP_GB', '0.8')) * 1e9 // per_req))
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
                if "--pushforward" in sys.argv:   # + the element's Piola map (fx_tabulate_batch_mapped): the basis ON the physical cells
                    def run():
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(5):
                            el.tabulate_batch(order, pts, verts=verts, out=out, pushforward=True)
                        e1.record()
                        torch.cuda.synchronize()
                        return e0.elapsed_time(e1) / 5
                    run()
                    t = statistics.median(run() for _ in range(3))
                else:
                    t = statistics.median(ps.time_tabulate_batch(order, pts, verts, out, 5) for _ in range(3))
                nexp = ps.coeffs.shape[-1] if hasattr(ps, "coeffs") else 0
                frac = per_req * nreq / t / 1e6 / 80
                kern = ps.kernel_name(order, nreq, npts, has_verts=verts is not None)
                rows.append((frac, f"{fam:22s} sd{sd} k{deg} order {order} npts {npts:3d} rows {int(np.prod(shape1[2:-1])):4d}: "
                                   f"{t*1e3:8.1f} us {nreq/t/1e3:9.1f} M/s {frac:5.1f} % HBM  {kern}"))
                print(rows[-1][1], flush=True)
                del pts, out
print("\n-- slowest 25 --")
for frac, line in sorted(rows)[:25]:
    print(line)
fr = np.array(sorted(r[0] for r in rows))
if len(fr):
    print(f"\n{len(fr)} shapes: geometric mean {np.exp(np.mean(np.log(fr))):.1f} % of the HBM peak, median {np.median(fr):.1f}, "
          f"quartiles {np.percentile(fr, 25):.1f} / {np.percentile(fr, 75):.1f}, min {fr[0]:.1f}, max {fr[-1]:.1f}")
