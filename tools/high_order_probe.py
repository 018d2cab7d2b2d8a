import os, sys, statistics
sys.path.insert(0, os.getcwd())
import numpy as np, torch, fiat_amd, bench
from fiat_amd import runtime
for fam, sd, deg, npts in (("Lagrange", 3, 3, 23), ("Lagrange", 2, 2, 6), ("Lagrange", 3, 2, 11), ("DiscontinuousLagrange", 3, 4, 23), ("Nedelec", 3, 2, 23)):
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg); ps = el.device_polyset()
    for order in (2, 3, 4):
        for use_verts in (False, True):
            shape1 = ps.out_shape(order, 1, npts)
            per_req = 8 * (npts * sd + int(np.prod(shape1[1:])))
            nreq = int(0.8e9 // per_req)
            pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
            verts = None
            if use_verts:
                rng = np.random.default_rng(3)
                A = torch.as_tensor(np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))).cuda()
                b = torch.as_tensor(rng.standard_normal((nreq, 1, sd))).cuda()
                ref = torch.as_tensor(np.array(fiat_amd.ufc_simplex(sd).get_vertices(), dtype=float)).cuda()
                verts = (torch.einsum("vd,red->rve", ref, A) + b).contiguous()
                pts = (torch.einsum("rpd,red->rpe", pts, A) + b).contiguous()
            out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
            def run():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): el.tabulate_batch(order, pts, verts=verts, out=out)
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 5
            run()
            t = statistics.median(run() for _ in range(3))
            print(f"{fam:22s} sd{sd} k{deg} order {order} verts {int(use_verts)} npts {npts}: {nreq:7d} req {t*1e3:8.1f} us {per_req*nreq/t/1e6/80:5.1f} % HBM", flush=True)
            del pts, out
