import os, sys, statistics
sys.path.insert(0, os.getcwd())
import numpy as np, torch, fiat_amd
from fiat_amd import runtime
ctx = runtime.Context.get()
# python tools/shared_probe.py ["family,sd,degree,points,order;..."]: one point set in many cells under the shared-kernel policies
# (points < 0: the default rule of degree -points)
SHAPES = (("Lagrange", 3, 3, -6, 1), ("Lagrange", 3, 2, -4, 1), ("Lagrange", 2, 2, -4, 1), ("Lagrange", 3, 1, -2, 1), ("Lagrange", 3, 3, -6, 0))
if len(sys.argv) > 1:
    SHAPES = [(f, int(a), int(b), int(c), int(d)) for f, a, b, c, d in (x.split(",") for x in sys.argv[1].split(";"))]
for fam, sd, deg, qd, order in SHAPES:
    cell = fiat_amd.ufc_simplex(sd)
    el = getattr(fiat_amd, fam)(cell, deg); ps = el.device_polyset()
    if qd < 0:
        rule = torch.as_tensor(np.asarray(fiat_amd.create_quadrature(cell, -qd).get_points())).cuda()
    else:
        e = np.random.default_rng(6).exponential(size=(qd, sd + 1))
        rule = torch.as_tensor((e / e.sum(axis=-1, keepdims=True))[:, 1:].copy()).cuda()
    npts = rule.shape[0]
    per_req = 8 * ((sd + 1) * sd + int(np.prod(ps.out_shape(order, 1, npts)[1:])))
    nreq = int(1.5e9 // per_req)
    rng = np.random.default_rng(3)
    A = torch.as_tensor(np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))).cuda()
    ref = torch.as_tensor(np.array(cell.get_vertices(), dtype=float)).cuda()
    verts = (torch.einsum("vd,red->rve", ref, A) + torch.as_tensor(rng.standard_normal((nreq, 1, sd))).cuda()).contiguous()
    out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
    res = []
    for pol in ((), ("no_shared_wave",), ("no_shared_wave", "no_shared_reg")):
        ctx.set_policy(*pol)
        def run():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): el.tabulate_cells(order, rule, verts, out=out)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 10
        run()
        t = statistics.median(run() for _ in range(3))
        res.append(f"{'+'.join(pol) or 'default':30s} {t*1e3:7.1f} us {per_req*nreq/t/1e6/80:5.1f} %")
    ctx.set_policy()
    print(f"{fam} sd{sd} k{deg} order {order} npts {npts} nreq {nreq}: " + " | ".join(res), flush=True)
