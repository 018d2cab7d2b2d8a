#!/usr/bin/env python3
"""Same-process, interleaved timing of the kernel families that can serve one workload (measurement tooling):
python tools/policy_ab.py n2tet rt2tet [--verts]  -> median us per launch under each kernel-selection policy."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fiat_amd import runtime

POLICIES = [(), ("no_fixed",), ("no_fixed", "no_stacked"), ("no_fixed", "no_stacked", "no_coop")]
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["n2tet", "rt2tet"]
for name in names:
    el, sd, deg, order, npts, batch = bench.build_element(name)
    ps = el.device_polyset()
    pts = torch.as_tensor(bench.synth_points(sd, batch, npts, 2)).cuda()
    out = torch.empty(ps.out_shape(order, batch, npts), dtype=torch.float64, device="cuda")
    byts = 8 * (npts * sd + int(np.prod(out.shape[1:]))) * batch
    ctx = runtime.Context.get()
    times = {p: [] for p in POLICIES}
    kern = {}
    for rnd in range(5):
        for p in POLICIES:
            ctx.set_policy(*p)
            kern[p] = ps.kernel_name(order, batch, npts)
            times[p].append(ps.time_tabulate_batch(order, pts, None, out, 20))
    ctx.set_policy()
    for p in POLICIES:
        med = statistics.median(times[p])
        print(f"{name:8s} {'+'.join(p) or 'default':32s} {kern[p]:34s} {med*1e3:8.1f} us  {byts/med/1e6/80:5.1f} % of 8 TB/s")
