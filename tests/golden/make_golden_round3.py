"""Golden vectors of round 3, generated from the *unmodified reference* in the build container:

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 python -B tests/golden/make_golden_round3.py

* ``ka_*``: the reference's primary known-answer test of the Dubiner recurrence (test/FIAT/unit/test_polynomial.py:34-84):
  expansion sets of degree 10 on the default interval / triangle / tetrahedron at the rational lattice points, what
  ``ExpansionSet.tabulate`` returns there AND the closed-form Jacobi products the reference test compares with (SymPy);
* ``hi_*``: degrees 7, 8 and 10 (the generic kernel's range) on the UFC cells, variants None / bubble, orders 0-2, at
  random points;
* ``pc_*``: derivative orders 3 and 4 of elements built by the reference ON physical cells (incl. a negatively oriented
  one), tabulated at physical points -- what ``tabulate_batch(order, points, verts=cells)`` must return for the affine
  families; for the non-affine cases (RT3 triangle, the raw 1-D set) the reference-cell tables of orders 3 and 4;
* ``gls_*``: TracelessTensorPolynomialSet (FIAT/polynomial_set.py:252-282) and GopalakrishnanLedererSchoberlSecondKind
  (FIAT/gopalakrishnan_lederer_schoberl.py:9-71): coefficients, entity dofs, order-1 tables, and the element built on a
  physical cell.
Plain numbers only."""
import json
import os

import numpy as np
import sympy

import FIAT
from FIAT import expansions, polynomial_set, reference_element
from FIAT.polynomial_set import mis
from FIAT.reference_element import UFCSimplex

HERE = os.path.dirname(os.path.abspath(__file__))


def simplex_points(rng, sd, n):
    e = rng.exponential(size=(n, sd + 1))
    return (e / e.sum(axis=1, keepdims=True))[:, 1:].copy()


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for k in range(order + 1) for a in mis(sd, k)])


def physical(sd, verts):
    ref = FIAT.ufc_simplex(sd)
    return UFCSimplex(ref.get_shape(), tuple(map(tuple, verts)), ref.get_topology())


def known_answer(dim, degree):
    """The arrays of test_polynomial.py:34-84: points, U.tabulate(degree, points), closed-form values."""
    cell = reference_element.default_simplex(dim)
    U = expansions.ExpansionSet(cell)
    dpoints, rpoints = [], []
    numpyoints, interior = 4, 1
    for alpha in reference_element.lattice_iter(interior, numpyoints + 1 - interior, dim):
        dpoints.append(tuple(2 * np.array(alpha, dtype="d") / numpyoints - 1))
        rpoints.append(tuple(2 * sympy.Rational(a, numpyoints) - 1 for a in alpha))
    Uvals = U.tabulate(degree, dpoints)
    idx = (lambda p: p, expansions.morton_index2, expansions.morton_index3)[dim - 1]
    eta = sympy.DeferredVector("eta")
    half = sympy.Rational(1, 2)

    def duffy(pt):
        if len(pt) == 1:
            return pt
        if len(pt) == 2:
            return 2 * (1 + pt[0]) / (1 - pt[1]) - 1, pt[1]
        return 2 * (1 + pt[0]) / (-pt[1] - pt[2]) - 1, 2 * (1 + pt[1]) / (1 - pt[2]) - 1, pt[2]

    def basis(p, q=0, r=0):
        f = sympy.jacobi(p, 0, 0, eta[0]) * sympy.sqrt(half + p)
        if dim >= 2:
            f *= sympy.jacobi(q, 2 * p + 1, 0, eta[1]) * ((1 - eta[1]) / 2) ** p * sympy.sqrt(1 + p + q)
        if dim >= 3:
            f *= sympy.jacobi(r, 2 * p + 2 * q + 2, 0, eta[2]) * ((1 - eta[2]) / 2) ** (p + q) * sympy.sqrt(1 + half + p + q + r)
        return f

    exact = np.zeros_like(np.asarray(Uvals))
    for i in range(degree + 1):
        for indices in mis(dim, i):
            phi = basis(*indices)
            exact[idx(*indices)] = [float(phi.subs(dict(zip(eta, duffy(r))))) for r in rpoints]
    assert np.allclose(Uvals, exact, atol=1e-14)        # the reference's own assertion
    return np.array(dpoints), np.asarray(Uvals), exact, np.array(cell.get_vertices(), dtype=float)


def main():
    rng = np.random.default_rng(303)
    out = {}
    # ---- degree-10 known answers -------------------------------------------------------------------------
    for dim in (1, 2, 3):
        pts, vals, exact, verts = known_answer(dim, 10)
        out[f"ka_sd{dim}_pts"], out[f"ka_sd{dim}_tab"], out[f"ka_sd{dim}_exact"], out[f"ka_sd{dim}_verts"] = pts, vals, exact, verts
        # derivatives at the same points (orders 1, 2) from the reference's recurrence
        U = expansions.ExpansionSet(reference_element.default_simplex(dim))
        out[f"ka_sd{dim}_jet2"] = stack(U._tabulate(10, pts, order=2), dim, 2)
    # ---- degrees 7, 8, 10 on the UFC cells ---------------------------------------------------------------
    for sd in (1, 2, 3):
        cell = FIAT.ufc_simplex(sd)
        pts = simplex_points(rng, sd, 9)
        out[f"hi_sd{sd}_pts"] = pts
        for variant in (None, "bubble"):
            U = expansions.ExpansionSet(cell, variant=variant)
            for n in (7, 8, 10):
                if sd == 3 and n == 10 and variant == "bubble":
                    continue
                out[f"hi_sd{sd}_{variant}_n{n}"] = stack(U._tabulate(n, pts, order=2), sd, 2)
    # ---- orders 3 and 4 on physical cells ----------------------------------------------------------------
    cases = [("p4tet", 3, lambda c: FIAT.Lagrange(c, 4), True), ("dg5tet", 3, lambda c: FIAT.DiscontinuousLagrange(c, 5), True),
             ("p5tri", 2, lambda c: FIAT.Lagrange(c, 5), True), ("rt3tri", 2, lambda c: FIAT.RaviartThomas(c, 3), False),
             ("on6int", 1, lambda c: polynomial_set.ONPolynomialSet(c, 6), False)]
    for name, sd, make, rebuild in cases:
        ref = np.array(FIAT.ufc_simplex(sd).get_vertices(), dtype=float)
        ncell, npts = 3, 11
        A = np.eye(sd) + 0.25 * rng.standard_normal((ncell, sd, sd))
        A[-1, :, 0] *= -1.0                                          # one negatively oriented cell
        verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((ncell, 1, sd))
        e = rng.exponential(size=(ncell, npts, sd + 1))
        bary = e / e.sum(-1, keepdims=True)
        pts, ref_pts = np.einsum("rpv,rvd->rpd", bary, verts), np.einsum("rpv,vd->rpd", bary, ref)
        out[f"pc_{name}_verts"], out[f"pc_{name}_pts"], out[f"pc_{name}_refpts"] = verts, pts, ref_pts
        base = make(FIAT.ufc_simplex(sd))
        for order in (3, 4):
            for r in range(ncell):
                if rebuild:
                    out[f"pc_{name}_o{order}_phys{r}"] = stack(make(physical(sd, verts[r])).tabulate(order, pts[r]), sd, order)
                else:
                    tab = base.tabulate(order, ref_pts[r]) if hasattr(base, "dual_basis") else base.tabulate(ref_pts[r], order)
                    out[f"pc_{name}_o{order}_ref{r}"] = stack(tab, sd, order)
    # ---- traceless tensors and GLS -----------------------------------------------------------------------
    for sd, k in ((2, 0), (2, 1), (2, 2), (3, 0), (3, 1)):
        cell = FIAT.ufc_simplex(sd)
        key = f"gls_sd{sd}_k{k}"
        P = polynomial_set.TracelessTensorPolynomialSet(cell, k)
        out[key + "_space"] = P.get_coeffs()
        el = FIAT.GopalakrishnanLedererSchoberlSecondKind(cell, k)
        out[key + "_coeffs"] = el.get_coeffs()
        out[key + "_entity_dofs"] = np.array(json.dumps({str(d): {str(i): list(map(int, v)) for i, v in ents.items()}
                                                         for d, ents in el.entity_dofs().items()}))
        out[key + "_mapping"] = np.array(el.mapping()[0])
        pts = simplex_points(rng, sd, 7)
        out[key + "_pts"] = pts
        out[key + "_tab"] = stack(el.tabulate(1, pts), sd, 1)
        # the same element built directly on a physical cell, at the images of the same points
        ref = np.array(cell.get_vertices(), dtype=float)
        A = np.eye(sd) + 0.2 * rng.standard_normal((sd, sd))
        verts = ref @ A.T + rng.standard_normal((1, sd))
        bary = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)
        ppts = bary @ verts
        out[key + "_phys_verts"], out[key + "_phys_pts"] = verts, ppts
        out[key + "_phys_tab"] = stack(FIAT.GopalakrishnanLedererSchoberlSecondKind(physical(sd, verts), k).tabulate(1, ppts), sd, 1)

    path = os.path.join(HERE, "round3.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
