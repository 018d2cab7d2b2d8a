import os, sys, statistics, math
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from fiat_amd import runtime
ctx = runtime.Context.get()
ctx.set_policy("no_fixed", "no_small", "no_stacked", "no_coop")
rng = np.random.default_rng(0)
for sd, n, rows, npts in ((3, 3, 20, 23), (3, 2, 45, 11), (2, 4, 15, 16), (3, 4, 35, 44), (3, 1, 36, 4)):
    nexp = math.comb(n + sd, sd)
    ps = runtime.SimplexPolySet(sd, n, coeffs=rng.standard_normal((rows, nexp)))
    for order in (0, 1, 2):
        per_req = 8 * (npts * sd + int(np.prod(ps.out_shape(order, 1, npts)[1:])))
        nreq = int(0.8e9 // per_req)
        pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, 1)).cuda()
        out = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
        t = statistics.median(ps.time_tabulate_batch(order, pts, None, out, 5) for _ in range(3))
        print(f"dbg {os.environ.get('FIAT_AMD_DEBUG','0')} sd{sd} n{n} rows {rows} order {order} npts {npts:3d}: {t*1e3:8.1f} us {per_req*nreq/t/1e6/80:5.1f} %  {ps.kernel_name(order, nreq, npts)[9:]}", flush=True)
        del pts, out
