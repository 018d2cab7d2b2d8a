import os, sys, statistics
sys.path.insert(0, os.getcwd())
import numpy as np, torch, fiat_amd, bench
from fiat_amd import runtime
for fam, sd, deg, npts, order in (("Lagrange", 2, 3, 12, 0), ("Lagrange", 3, 2, 11, 0), ("Lagrange", 2, 1, 3, 1), ("Lagrange", 2, 2, 6, 0)):
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg); ps = el.device_polyset()
    per_req = 8 * (npts * sd + int(np.prod(ps.out_shape(order, 1, npts)[1:])))
    nreq = int(min(4e6, 0.8e9 // per_req))
    pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
    out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
    t = statistics.median(ps.time_tabulate_batch(order, pts, None, out, 5) for _ in range(3))
    print(f"debug {os.environ.get('FIAT_AMD_DEBUG','0')} {fam} sd{sd} k{deg} order {order} npts {npts}: {t*1e3:7.1f} us {per_req*nreq/t/1e6/80:5.1f} %  {ps.kernel_name(order, nreq, npts)}", flush=True)
