#!/usr/bin/env python3
"""Timing of general tensor products (factor tables x table_outer_kernel): prisms P2 x P1, P3 x P2 and an H(div)-style
RT1 x DG0 product, order 1 (measurement tooling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd as fa

T, I = fa.ufc_simplex(2), fa.ufc_simplex(1)
cases = {"P2tri x P1": fa.TensorProductElement(fa.Lagrange(T, 2), fa.Lagrange(I, 1)),
         "P3tri x P2": fa.TensorProductElement(fa.Lagrange(T, 3), fa.Lagrange(I, 2)),
         "RT1tri x DG0": fa.TensorProductElement(fa.RaviartThomas(T, 1), fa.DiscontinuousLagrange(I, 0))}
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 18
for name, el in cases.items():
    ndof = el.space_dimension()
    vdim = int(np.prod(el.value_shape() or (1,)))
    per = 8 * (3 * npts + 4 * ndof * vdim * npts)
    nreq = int(1.0e9 // per)
    rng = np.random.default_rng(0)
    e = rng.exponential(size=(nreq, npts, 3))
    tri = (e / e.sum(-1, keepdims=True))[..., 1:]
    pts = torch.as_tensor(np.concatenate([tri, rng.uniform(0, 1, size=(nreq, npts, 1))], axis=-1)).cuda()
    out = el.tabulate_batch(1, pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        el.tabulate_batch(1, pts, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:14s} {nreq:8d} requests x {npts} points: {ms*1e3:8.1f} us  {nreq/ms/1e3:8.1f} M req/s  {per*nreq/ms/1e6:6.0f} GB/s ({per*nreq/ms/1e6/80:.0f} % of 8 TB/s)")
