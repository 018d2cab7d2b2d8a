#!/usr/bin/env python3
"""Which kernel serves which realistic request shape, and how close to the HBM / fp64 rooflines it runs (measurement tooling):
families x cell x degree x derivative order, points = the default quadrature rule of degree 2 * degree (what a mass /
stiffness assembly asks for).  python tools/coverage_map.py [--verts [--pushforward]] [--order K] [--policy no_small,no_stacked] [--only "Lagrange sd3"] [--qdeg-offset K] [--audit]
(--audit: every shape also under each kernel-selection policy; lists the shapes another family serves > 7 % faster)"""
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
qoff = int(sys.argv[sys.argv.index("--qdeg-offset") + 1]) if "--qdeg-offset" in sys.argv else 0   # rule degree 2 * degree + this
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
rows, audit = [], []
AUDIT = [("no_stacked",), ("no_small",), ("no_stacked", "no_small"), ("no_fixed",), ("no_stacked_mix",), ("stacked_small",), ("no_wg",),
         ("wg_small",), ("wg_small", "no_fixed", "no_small")]
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
            npts = len(fiat_amd.create_quadrature(cell, max(1, 2 * deg + qoff)).get_points())
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
                if "--audit" in sys.argv:   # is the planner's choice the fastest family here?
                    ctx, alt = runtime.Context.get(), {}
                    for rnd in range(3):                      # interleaved, the default among them: clock and placement drift cancel
                        for pol in [()] + AUDIT:
                            ctx.set_policy(*pol)
                            try:
                                pf = "--pushforward" in sys.argv
                                tt = run() if pf else ps.time_tabulate_batch(order, pts, verts, out, 5)
                                alt.setdefault(pol, ([], ps.kernel_name(order, nreq, npts, has_verts=verts is not None, instance=True,
                                                                        mapping=el.mapping()[0] if pf and el.mapping()[0] != "affine" else None)))[0].append(tt)
                            except Exception as e:
                                alt[pol] = ([float("inf")], str(e)[:40])
                    alt = {q: (statistics.median(v[0]), v[1]) for q, v in alt.items()}
                    t = alt.pop(())[0]
                    ctx.set_policy()
                    best = min(alt, key=lambda q: alt[q][0])
                    if alt[best][0] < 0.93 * t:
                        audit.append(f"{fam} sd{sd} k{deg} order {order} npts {npts}: default {t*1e3:.1f} us, "
                                     f"{'+'.join(best)} {alt[best][0]*1e3:.1f} us ({alt[best][1]})")
                        print("   AUDIT " + audit[-1] + "\n      " + "  ".join(f"{'+'.join(q)}={v[0]*1e3:.1f}" for q, v in alt.items()), flush=True)
                # both rooflines: algorithmic bytes against the 8 TB/s HBM peak, contraction flops (2 x stacked rows x expansion
                # members x points) against the fp64 peak of bench.py; the BINDING one is the larger time.  Degree >= 5 on
                # tetrahedra is fp64-bound (2 nexp / 8 flop per output byte against a machine balance of 9.8).
                import math
                nexp = math.comb(el.get_nodal_basis().get_embedded_degree() + sd, sd)
                flops = 2.0 * int(np.prod(shape1[1:-1])) * nexp * npts * nreq
                frac = per_req * nreq / t / 1e6 / 80
                ffrac = flops / t / 1e9 / bench.F64_PEAK_TFLOPS * 100          # (t in ms)
                bind = max(frac, ffrac)
                resident = per_req * nreq < 0.5e9       # the whole output stays in the 256 MB Infinity Cache: not an HBM measurement
                kern = ps.kernel_name(order, nreq, npts, has_verts=verts is not None)
                rows.append((frac, f"{fam:22s} sd{sd} k{deg} order {order} npts {npts:3d} rows {int(np.prod(shape1[2:-1])):4d}: "
                                   f"{t*1e3:8.1f} us {nreq/t/1e3:9.1f} M/s {frac:5.1f} % HBM {ffrac:5.1f} % fp64 -> {bind:5.1f} % {'fp64' if ffrac > frac else 'HBM '}"
                                   f"{' (cache-resident)' if resident else ''}  {kern}", bind, resident))
                print(rows[-1][1], flush=True)
                del pts, out
if audit:
    print(f"\n-- audit: {len(audit)} shapes where another kernel family beats the planner's choice by > 7 % --")
    print("\n".join(audit))
print("\n-- slowest 25 (by the binding roofline) --")
for r in sorted(rows, key=lambda r: r[2])[:25]:
    print(r[1])
res = [r for r in rows if r[3]]
if res:
    print(f"\n{len(res)} cache-resident shapes (outputs below 0.5 GB: their 'HBM' figure measures the Infinity Cache), excluded from the means:")
    for r in res:
        print("  " + r[1])
for label, k in (("HBM peak", 0), ("binding roofline (HBM or fp64)", 2)):
    fr = np.array(sorted(r[k] for r in rows if not r[3]))
    if len(fr):
        print(f"\n{len(fr)} shapes, % of the {label}: geometric mean {np.exp(np.mean(np.log(fr))):.1f}, median {np.median(fr):.1f}, "
              f"quartiles {np.percentile(fr, 25):.1f} / {np.percentile(fr, 75):.1f}, min {fr[0]:.1f}, max {fr[-1]:.1f}")
