#!/usr/bin/env python3
"""The quadrature-rule case over realistic shapes (measurement tooling): ONE rule of degree 2 * degree on the reference cell,
pushed forward to many physical cells (``tabulate_cells`` = fx_tabulate_batch_shared, with the element's Piola map).
Bytes counted: per-request cell + tables.  python tools/coverage_map_cells.py [--order K] [--audit]
(--audit: each shape also under the no_shared_wave / no_shared_reg policies; lists shapes another kernel serves > 7 % faster)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd
from fiat_amd import runtime
FAMS = [("Lagrange", range(1, 7)), ("DiscontinuousLagrange", range(0, 7)), ("Nedelec", range(1, 5)), ("RaviartThomas", range(1, 5)),
        ("BrezziDouglasMarini", range(1, 4)), ("NedelecSecondKind", range(1, 4))]
orders = [int(sys.argv[sys.argv.index("--order") + 1])] if "--order" in sys.argv else [0, 1, 2]
rows = []
for sd in (2, 3):
    cell = fiat_amd.ufc_simplex(sd)
    ref = np.array(cell.get_vertices(), dtype=float)
    for fam, degs in FAMS:
        for deg in degs:
            if sd == 3 and fam in ("Nedelec", "RaviartThomas") and deg > 3:
                continue
            el = getattr(fiat_amd, fam)(cell, deg)
            ps = el.device_polyset()
            rule = torch.as_tensor(np.asarray(fiat_amd.create_quadrature(cell, max(1, 2 * deg)).get_points())).cuda()
            npts = rule.shape[0]
            for order in orders:
                shape1 = ps.out_shape(order, 1, npts)
                per_req = 8 * ((sd + 1) * sd + int(np.prod(shape1[1:])))
                nreq = int(min(2_000_000, 0.8e9 // per_req))
                rng = np.random.default_rng(3)
                A = torch.as_tensor(np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))).cuda()
                b = torch.as_tensor(rng.standard_normal((nreq, 1, sd))).cuda()
                verts = (torch.einsum("vd,red->rve", torch.as_tensor(ref).cuda(), A) + b).contiguous()
                out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
                fn = lambda: el.tabulate_cells(order, rule, verts, out=out)
                for _ in range(2): fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5): fn()
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 5)
                t = statistics.median(ts)
                if "--audit" in sys.argv:   # the other kernels of fx_tabulate_batch_shared, interleaved with the default
                    ctx, alt = runtime.Context.get(), {}
                    for rnd in range(3):
                        for pol in [(), ("no_shared_wave",), ("no_shared_reg",), ("no_shared_wave", "no_shared_reg")]:
                            ctx.set_policy(*pol)
                            fn()
                            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(5): fn()
                            e1.record(); torch.cuda.synchronize()
                            alt.setdefault(pol, []).append(e0.elapsed_time(e1) / 5)
                    ctx.set_policy()
                    alt = {q: statistics.median(v) for q, v in alt.items()}
                    t = alt.pop(())
                    best = min(alt, key=alt.get)
                    if alt[best] < 0.93 * t:
                        print(f"   AUDIT {fam} sd{sd} k{deg} order {order} npts {npts}: default {t*1e3:.1f} us; " +
                              "  ".join(f"{'+'.join(q)}={v*1e3:.1f}" for q, v in alt.items()), flush=True)
                frac = per_req * nreq / t / 1e6 / 80
                rows.append((frac, f"{fam:22s} sd{sd} k{deg} order {order} npts {npts:3d} rows {int(np.prod(shape1[2:-1])):4d} {el.mapping()[0][:12]:12s}: "
                                   f"{t*1e3:8.1f} us {nreq/t/1e3:9.1f} M/s {frac:5.1f} % HBM"))
                print(rows[-1][1], flush=True)
                del verts, out, A, b
print("\n-- slowest 25 --")
for frac, line in sorted(rows)[:25]:
    print(line)
import statistics as st
print("geo-mean %.1f %%" % st.geometric_mean([r[0] for r in rows]))
