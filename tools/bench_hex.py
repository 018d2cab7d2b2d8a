#!/usr/bin/env python3
"""C5 timing: P4 x P4 x P4 hexahedron, order 1, 5^3 tensor grid per request (measurement tooling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd
from oracle import fiat_oracle as fo

nreq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
I = fiat_amd.ufc_simplex(1)
P4 = fiat_amd.Lagrange(I, 4)
hexel = fiat_amd.TensorProductElement(fiat_amd.TensorProductElement(P4, P4), P4)
rng = np.random.default_rng(5)
grid = np.sort(rng.uniform(0, 1, size=(nreq, 3, 5)), axis=2)
gd = torch.as_tensor(grid).cuda()
out = torch.empty((nreq, 4, 125, 125), dtype=torch.float64, device="cuda")
pts = None
for mode in ("grid", "points"):
    if mode == "points":
        g = grid[:2000]
        pts = torch.as_tensor(np.stack([np.array([[x, y, z] for x in gg[0] for y in gg[1] for z in gg[2]]) for gg in g])).cuda()
        n = 2000
        o = out[:n]
        fn = lambda: hexel.tabulate_batch(1, pts, out=o)
    else:
        n = nreq
        fn = lambda: hexel.tabulate_batch(1, gd, out=out, grid=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = n * 4 * 125 * 125 * 8
    print(f"hex {mode}: {n} requests, {ms:.3f} ms -> {n/ms*1e3:.3g} tab/s, {by/ms/1e6:.0f} GB/s ({by/ms/1e6/8000:.1%} of 8 TB/s)")
ref = fo.hex_lagrange_tabulate(np.array(P4.get_nodal_basis().get_expansion_set().x), 1,
                               np.array([[x, y, z] for x in grid[7][0] for y in grid[7][1] for z in grid[7][2]]))
got = out[7].cpu().numpy()
err = max(np.abs(got[t] - ref[a]).max() for t, a in enumerate(fo.jet_indices(3, 1)))
print("max abs err vs oracle", err)
