#!/usr/bin/env python3
"""Throughput of tabulation + Piola push-forward for N2 / RT2 tetrahedra,
25 000 requests x 23 points, per-request cells (measurement tooling)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd, bench

def timed(fn, reps=10, rounds=5):
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)

nreq, npts = 25000, 23
rng = np.random.default_rng(5)
ref = np.array(fiat_amd.ufc_simplex(3).get_vertices(), dtype=float)
A = np.eye(3) + 0.1 * rng.standard_normal((nreq, 3, 3))
b = rng.standard_normal((nreq, 1, 3))
verts = torch.as_tensor(np.einsum("vd,red->rve", ref, A) + b).cuda()
pts = torch.as_tensor(np.einsum("rpd,red->rpe", bench.synth_points(3, nreq, npts, 3), A) + b).cuda()
for fam in ("Nedelec", "RaviartThomas"):
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(3), 2)
    ps = el.device_polyset()
    out = torch.empty(ps.out_shape(1, nreq, npts), dtype=torch.float64, device="cuda")
    by = out.numel() * 8
    t_ref = timed(lambda: ps.tabulate_batch(1, pts, out=out))
    t_cell = timed(lambda: ps.tabulate_batch(1, pts, verts=verts, out=out))
    t_push = timed(lambda: el.tabulate_batch(1, pts, verts=verts, out=out, pushforward=True))
    print(f"{fam:14s} degree 2: reference cell {t_ref*1e3:7.1f} us | physical cells {t_cell*1e3:7.1f} us | + {el.mapping()[0]} "
          f"{t_push*1e3:7.1f} us -> {nreq/t_push*1e3:.3g} tab/s, {by/t_push/1e6:.0f} GB/s of tables "
          f"(push-forward pass alone {1e3*(t_push-t_cell):.1f} us = {2*by/(t_push-t_cell)/1e6:.0f} GB/s read+write)")
