#!/usr/bin/env python3
"""Range-checked run of the in-kernel table-mixing instances of the stacked kernel (FX_DBG=1024 build):
FIAT_AMD_LIB=.../libfiat_amd_dbg1024.so FIAT_AMD_STACKED_MIX=1 python tools/mix_probe.py <family> <degree> <npts> <nreq>.
Every global access is range-checked in that build (redirected + reported instead of faulting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fiat_amd, bench
from oracle import c_oracle, fiat_oracle as fo
fam, deg, npts, nreq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd, order = 3, 1
el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
ps = el.device_polyset()
rng = np.random.default_rng(3)
pts = bench.synth_points(sd, nreq, npts, 1)
A = np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))
b = rng.standard_normal((nreq, 1, sd))
verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[sd], A) + b
pts = np.einsum("rpd,red->rpe", pts, A) + b
print(ps.kernel_name(order, nreq, npts, has_verts=True), flush=True)
out = ps.tabulate_batch(order, pts, verts=verts)
torch.cuda.synchronize()
print("launch done", flush=True)
out = out.cpu().numpy()
sel = np.unique(np.concatenate([np.arange(0, nreq, max(1, nreq // 3000)), [nreq - 1]]))
n = el.get_nodal_basis().get_embedded_degree()
ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[sd], n, el.get_coeffs(), order, pts[sel], verts=verts[sel], scale=el._expansion_scale,
                              variant=el._expansion_variant).reshape(out[sel].shape)
err = np.abs(out[sel] - ref).max() / max(1.0, np.abs(ref).max())
print("max rel err on %d sampled requests: %.3e" % (len(sel), err), flush=True)
dpts, dverts = torch.as_tensor(pts).cuda(), torch.as_tensor(verts).cuda()
dout = torch.empty(ps.out_shape(order, nreq, npts), dtype=torch.float64, device="cuda")
t = min(ps.time_tabulate_batch(order, dpts, dverts, dout, 10) for _ in range(3))
print("%.1f us per launch, %.0f GB/s of tables" % (t * 1e3, dout.numel() * 8 / t / 1e6), flush=True)
