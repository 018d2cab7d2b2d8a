#!/usr/bin/env python3
"""Randomised cross-check of the kernel families (measurement / debugging tooling): for random elements, orders, point counts,
batch sizes and optional per-request cells, the default kernel selection must agree with the generic kernel
(policy no_fixed + no_small + no_stacked + no_coop).  python tools/fuzz_policies.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fiat_amd as fa
from fiat_amd import runtime

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = runtime.Context.get()
FAMS = [("Lagrange", 1, 6), ("DiscontinuousLagrange", 0, 6), ("Nedelec", 1, 4), ("RaviartThomas", 1, 4), ("BrezziDouglasMarini", 1, 3),
        ("NedelecSecondKind", 1, 3)]
cache = {}
t0, n, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    fam, lo, hi = FAMS[rng.integers(len(FAMS))]
    sd = int(rng.integers(2, 4))
    deg = int(rng.integers(lo, hi + 1))
    if fam in ("Nedelec", "RaviartThomas", "BrezziDouglasMarini", "NedelecSecondKind") and sd == 3 and deg > 3:
        deg = 3
    key = (fam, sd, deg)
    if key not in cache:
        cache[key] = getattr(fa, fam)(fa.ufc_simplex(sd), deg)
    el = cache[key]
    order = int(rng.integers(0, 3))
    npts = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 21, 23, 24, 25, 31, 32, 33, 40, 48, 49, 64, 65, 70, 122]))
    nreq = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 257, 1000, 4097]))
    rows = el.space_dimension() * int(np.prod(el.value_shape() or (1,)))
    if nreq * npts * rows * (1 + sd + sd * (sd + 1) // 2) * 8 > 2e9:
        nreq = 7
    e = rng.exponential(size=(nreq, npts, sd + 1))
    bary = e / e.sum(-1, keepdims=True)
    ref = np.array(fa.ufc_simplex(sd).get_vertices(), dtype=float)
    verts = None
    if rng.random() < 0.5:
        A = np.eye(sd) + 0.2 * rng.standard_normal((nreq, sd, sd))
        verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((nreq, 1, sd))
        pts = np.einsum("rpv,rvd->rpd", bary, verts)
    else:
        pts = np.einsum("rpv,vd->rpd", bary, ref)
    # vector-valued elements on per-request cells: half of the time with their Piola map (fused where a kernel fuses it,
    # a pass of its own on the generic route)
    push = verts is not None and el.mapping()[0] != "affine" and rng.random() < 0.5
    ctx.set_policy()
    kern = el.device_polyset().kernel_name(order, nreq, npts, has_verts=verts is not None, instance=True,
                                           mapping=el.mapping()[0] if push else None)
    a = el.tabulate_batch(order, pts, verts=verts, pushforward=push).cpu().numpy()
    ctx.set_policy("no_fixed", "no_small", "no_stacked", "no_coop")
    b = el.tabulate_batch(order, pts, verts=verts, pushforward=push).cpu().numpy()
    ctx.set_policy()
    axes = tuple(range(2, a.ndim))
    err = float((np.abs(a - b).max(axis=axes) / np.maximum(1.0, np.abs(b).max(axis=axes))).max())
    worst = max(worst, err)
    n += 1
    if not np.isfinite(a).all() or err > 1e-9:
        print("MISMATCH", key, "order", order, "npts", npts, "nreq", nreq, "verts", verts is not None, "push", push, kern, err, flush=True)
# one rule in many cells (tabulate_cells: flat / wave / register-resident streaming kernels) against per-request points with the
# element's push-forward on the generic route
t2, ncells = time.time(), 0
while time.time() - t2 < budget / 4:
    fam, lo, hi = FAMS[rng.integers(len(FAMS))]
    sd = int(rng.integers(2, 4))
    deg = int(rng.integers(lo, min(hi, 4) + 1))
    if fam in ("Nedelec", "RaviartThomas", "BrezziDouglasMarini", "NedelecSecondKind") and sd == 3 and deg > 3:
        deg = 3
    key = (fam, sd, deg)
    if key not in cache:
        cache[key] = getattr(fa, fam)(fa.ufc_simplex(sd), deg)
    el = cache[key]
    order = int(rng.integers(0, 3))
    npts = int(rng.choice([1, 2, 3, 4, 6, 7, 11, 12, 16, 23, 24, 33]))
    nreq = int(rng.choice([1, 2, 63, 64, 65, 130, 1000, 4097]))
    e = rng.exponential(size=(npts, sd + 1))
    bary = e / e.sum(-1, keepdims=True)
    ref = np.array(fa.ufc_simplex(sd).get_vertices(), dtype=float)
    A = np.eye(sd) + 0.2 * rng.standard_normal((nreq, sd, sd))
    verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((nreq, 1, sd))
    ctx.set_policy()
    a = el.tabulate_cells(order, bary @ ref, verts).cpu().numpy()
    ctx.set_policy("no_fixed", "no_small", "no_stacked", "no_coop")
    b = el.tabulate_batch(order, np.einsum("pv,rvd->rpd", bary, verts), verts=verts, pushforward=True).cpu().numpy()
    ctx.set_policy()
    axes = tuple(range(2, a.ndim))
    err = float((np.abs(a - b).max(axis=axes) / np.maximum(1.0, np.abs(b).max(axis=axes))).max())
    worst = max(worst, err)
    ncells += 1
    if not np.isfinite(a).all() or err > 1e-9:
        print("MISMATCH cells", key, "order", order, "npts", npts, "nreq", nreq, err, flush=True)
# tensor products and prisms: lane-local / fused kernels against the per-request and general routes (policy no_small)
I = fa.ufc_simplex(1)
tp_cache = {}
t1, m = time.time(), 0
while time.time() - t1 < budget / 4:
    kind = rng.integers(3)
    order = int(rng.integers(0, 3))
    nreq = int(rng.choice([1, 2, 5, 63, 64, 65, 1000, 4097]))
    if kind < 2:                                   # quadrilateral / hexahedron of equal-degree Lagrange factors
        dim, k = int(rng.integers(2, 4)), int(rng.integers(1, 5))
        key = ("q", dim, k)
        if key not in tp_cache:
            P = fa.Lagrange(I, k)
            el = fa.TensorProductElement(P, P)
            tp_cache[key] = fa.TensorProductElement(el, P) if dim == 3 else el
        el = tp_cache[key]
        npts = int(rng.choice([1, 3, 4, 8, 9, 16, 27, 30, 64]))
        if npts * (k + 1) ** dim * 10 * 8 * nreq > 1e9:
            nreq = 5
        pts = rng.uniform(0, 1, size=(nreq, npts, dim))
        if rng.random() < 0.3:
            pts[0, 0, 0] = 1.0                     # a coordinate on a node
    else:                                          # prism
        fam, ka = [("Lagrange", 1), ("Lagrange", 2), ("Lagrange", 3), ("RaviartThomas", 1), ("Nedelec", 1), ("DiscontinuousLagrange", 1)][rng.integers(6)]
        kb = int(rng.integers(1, 4))
        key = ("p", fam, ka, kb)
        if key not in tp_cache:
            tp_cache[key] = fa.TensorProductElement(getattr(fa, fam)(fa.ufc_simplex(2), ka), fa.Lagrange(I, kb))
        el = tp_cache[key]
        npts = int(rng.choice([1, 2, 6, 7, 12, 18, 33, 64]))
        e = rng.exponential(size=(nreq, npts, 3))
        pts = np.concatenate([(e / e.sum(-1, keepdims=True))[..., 1:], rng.uniform(0, 1, size=(nreq, npts, 1))], axis=-1)
    ctx.set_policy()
    a = el.tabulate_batch(order, pts).cpu().numpy()
    ctx.set_policy("no_small")
    b = el.tabulate_batch(order, pts).cpu().numpy()
    ctx.set_policy()
    axes = tuple(range(2, a.ndim))
    err = float((np.abs(a - b).max(axis=axes) / np.maximum(1.0, np.abs(b).max(axis=axes))).max())
    worst = max(worst, err)
    m += 1
    if not np.isfinite(a).all() or err > 1e-9:
        print("MISMATCH tensor", key, "order", order, "npts", npts, "nreq", nreq, err, flush=True)
ctx.check()
print(f"{n} random cases + {ncells} one-rule-many-cells cases + {m} tensor / prism cases, worst relative difference {worst:.2e}")
