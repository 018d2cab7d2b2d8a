import os, sys, statistics, math
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench, fiat_amd
from fiat_amd import runtime
ctx = runtime.Context.get()
for fam, sd, deg, nps in (("Lagrange", 3, 4, (17, 23)), ("RaviartThomas", 3, 2, (11, 23)), ("Nedelec", 3, 3, (23,)), ("Lagrange", 2, 5, (25,))):
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg); ps = el.device_polyset()
    for npts in nps:
        for order in (0, 1, 2):
            per_req = 8 * (npts * sd + int(np.prod(ps.out_shape(order, 1, npts)[1:])))
            nreq = int(0.8e9 // per_req)
            pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
            out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
            res = []
            for pol in ((), ("no_stacked",)):
                ctx.set_policy(*pol)
                k = ps.kernel_name(order, nreq, npts).replace("fxk::tabulate_simplex_", "")
                t = statistics.median(ps.time_tabulate_batch(order, pts, None, out, 5) for _ in range(3))
                res.append(f"{k:8s} {per_req*nreq/t/1e6/80:5.1f} %")
            ctx.set_policy()
            print(f"{fam:14s} sd{sd} k{deg} order {order} npts {npts:3d}: " + "   ".join(res), flush=True)
            del pts, out
