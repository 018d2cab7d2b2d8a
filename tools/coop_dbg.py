import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fiat_amd import runtime
from oracle import fiat_oracle as fo, c_oracle
g = np.load("tests/golden/elements.npz")
co = g["c4_dg6tet_q6_coeffs"]
rng = np.random.default_rng(0)
e = rng.exponential(size=(3, 23, 4)); pts = (e / e.sum(-1, keepdims=True))[..., 1:].copy()
for order in (1, 2):
    ps = runtime.SimplexPolySet(3, 6, coeffs=co)
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 6, co, order, pts)
    out = ps.tabulate_batch(order, pts).cpu().numpy()
    err = np.abs(out - ref).max(axis=(0, 2, 3))
    print("order", order, "uniform per-table err", err)
    verts = np.tile(fo.UFC_SIMPLEX[3], (3, 1, 1))
    out = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    print("order", order, "verts   per-table err", np.abs(out - ref).max(axis=(0, 2, 3)))
    # identity coefficients: which members are wrong?
    es = runtime.SimplexPolySet(3, 6)
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 6, np.eye(84), order, pts)
    out = es.tabulate_batch(order, pts).cpu().numpy()
    bad = np.where(np.abs(out - ref).max(axis=(0, 1, 3)) > 1e-9)[0]
    print("order", order, "bad members (identity coeffs):", bad.tolist())
