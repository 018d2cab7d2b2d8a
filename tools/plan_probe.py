#!/usr/bin/env python3
"""Which kernel / stacked instance the planner picks (measurement tooling): python tools/plan_probe.py "Nedelec,2,3,12,1;..." [--policy a,b]
(family, sd, degree, points, order); printed for own cell, per-request cells, per-request cells + the element's Piola map."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiat_amd
from fiat_amd import runtime
if "--policy" in sys.argv:
    runtime.Context.get().set_policy(*sys.argv[sys.argv.index("--policy") + 1].split(","))
for spec in sys.argv[1].split(";"):
    fam, sd, deg, npts, order = spec.split(",")
    sd, deg, npts, order = int(sd), int(deg), int(npts), int(order)
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
    ps = el.device_polyset()
    m = el.mapping()[0]
    names = [ps.kernel_name(order, 100000, npts, instance=True), ps.kernel_name(order, 100000, npts, has_verts=True, instance=True)]
    if m != "affine":
        names.append(ps.kernel_name(order, 100000, npts, has_verts=True, instance=True, mapping=m))
    print(f"{spec:32s} " + " | ".join(n.replace("fxk::tabulate_simplex_", "") for n in names), flush=True)
