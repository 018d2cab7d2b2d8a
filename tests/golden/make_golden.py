"""Generate golden vectors from the *unmodified reference* (run in the build
container only; /root/reference does not travel to the GPU box).

    cd /root/repo && \
    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tests/golden/make_golden.py

Writes tests/golden/*.npz (plain numbers only: inputs + the reference's
outputs).  ``recursivenodes`` is absent from the image; the restatement under
oracle/restated_deps stands in for it (equispaced nodes and exact Gauss-Jacobi
rules only -- the in-scope variants).
"""
import os
import sys

import numpy as np

import FIAT
from FIAT import (DiscontinuousLagrange, Lagrange, Nedelec, RaviartThomas,
                  TensorProductElement, create_quadrature, ufc_simplex)
from FIAT import barycentric_interpolation, expansions, jacobi, reference_element
from FIAT.polynomial_set import mis
from FIAT.reference_element import (DefaultLine, DefaultTetrahedron, DefaultTriangle,
                                    UFCInterval, make_affine_mapping, make_lattice)

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_simplex_points(rng, sd, n):
    e = rng.exponential(size=(n, sd + 1))
    bary = e / e.sum(axis=1, keepdims=True)
    return bary[:, 1:].copy()


def jet_keys(sd, order):
    return [a for k in range(order + 1) for a in mis(sd, k)]


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for a in jet_keys(sd, order)])


class PhysicalSimplex(reference_element.UFCSimplex):
    pass


def physical_cell(sd, verts):
    base = ufc_simplex(sd)
    return reference_element.UFCSimplex(base.get_shape(), tuple(map(tuple, verts)), base.get_topology())


def main():
    rng = np.random.default_rng(20261003)
    out = {}

    # ---- a1: Jacobi tables ------------------------------------------------
    xs = np.linspace(-1.0, 1.0, 9)[:, None]
    for (a, b) in [(0, 0), (1, 1), (2, 2), (3, 0), (5, 1)]:
        out[f"jacobi_{a}_{b}"] = jacobi.eval_jacobi_batch(a, b, 7, xs)
        out[f"jacobi_deriv_{a}_{b}"] = jacobi.eval_jacobi_deriv_batch(a, b, 7, xs)
    out["jacobi_x"] = xs
    np.savez(os.path.join(HERE, "jacobi.npz"), **out)

    # ---- a5-a7: raw expansion tables ---------------------------------------
    out = {}
    cells = {1: [UFCInterval(), DefaultLine()],
             2: [ufc_simplex(2), DefaultTriangle()],
             3: [ufc_simplex(3), DefaultTetrahedron()]}
    for sd in (1, 2, 3):
        pts = rand_simplex_points(rng, sd, 7)
        out[f"pts_sd{sd}"] = pts
        for ci, cell in enumerate(cells[sd]):
            verts = np.array(cell.get_vertices())
            A, b = make_affine_mapping(ufc_simplex(sd).get_vertices(), verts)
            cpts = pts @ A.T + b            # same barycentric points on this cell
            out[f"verts_sd{sd}_c{ci}"] = verts
            out[f"cpts_sd{sd}_c{ci}"] = cpts
            for variant in (None, "bubble", "dual"):
                for n in (0, 1, 2, 3, 4, 6):
                    for order in (0, 1, 2, 3):
                        if order == 3 and n not in (3, 4):
                            continue
                        if variant == "bubble" and n == 0:
                            continue    # the reference itself raises (C0_basis needs n >= 1)
                        es = expansions.ExpansionSet(cell, variant=variant)
                        tab = es._tabulate(n, cpts, order)
                        out[f"exp_sd{sd}_c{ci}_{variant}_n{n}_o{order}"] = stack(tab, sd, order)
        # physical (random affine) cell, scale=1 bubble as Lagrange uses it
        verts = np.array(ufc_simplex(sd).get_vertices()) + rng.uniform(-0.2, 0.2, size=(sd + 1, sd))
        cell = physical_cell(sd, verts)
        A, b = make_affine_mapping(ufc_simplex(sd).get_vertices(), verts)
        cpts = pts @ A.T + b
        out[f"verts_sd{sd}_phys"] = verts
        out[f"cpts_sd{sd}_phys"] = cpts
        for variant, scale in ((None, None), ("bubble", 1)):
            es = expansions.ExpansionSet(cell, scale=scale, variant=variant)
            for n in (2, 3):
                tab = es._tabulate(n, cpts, 2)
                out[f"exp_sd{sd}_phys_{variant}_n{n}_o2"] = stack(tab, sd, 2)
        Am, bm = make_affine_mapping(verts, reference_element.default_simplex(sd).get_vertices())
        out[f"affine_A_sd{sd}"] = Am
        out[f"affine_b_sd{sd}"] = bm
    # single point in -> trailing axis dropped (test_fiat.py:659-668)
    es = expansions.ExpansionSet(ufc_simplex(2))
    out["single_point_tri_n2"] = es._tabulate(2, np.array([0.25, 0.5]), 0)[(0, 0)]
    np.savez_compressed(os.path.join(HERE, "expansion.npz"), **out)

    # ---- a8: barycentric Lagrange ------------------------------------------
    out = {}
    nodes = np.linspace(0.0, 1.0, 5)
    dmat, wts = barycentric_interpolation.make_dmat(nodes)
    pts = np.concatenate([rng.uniform(0, 1, size=6), [0.25, 1.0]])[:, None]
    tab = barycentric_interpolation.barycentric_interpolation(nodes, wts, dmat, pts, order=2)
    out["nodes"] = nodes
    out["dmat"] = dmat
    out["wts"] = wts
    out["pts"] = pts
    out["tab"] = np.stack([tab[(r,)] for r in range(3)])
    np.savez(os.path.join(HERE, "lagrange_line.npz"), **out)

    # ---- elements of the five BASELINE configs -----------------------------
    out = {}
    T2, T3 = ufc_simplex(2), ufc_simplex(3)

    def dump_element(tag, el, sd, order, pts):
        out[f"{tag}_coeffs"] = el.get_coeffs()
        out[f"{tag}_V"] = el.V
        out[f"{tag}_pts"] = pts
        tab = el.tabulate(order, pts)
        out[f"{tag}_tab"] = stack(tab, sd, order)

    # C1: P1 triangle, order 1, 3-pt rule
    Q = create_quadrature(T2, 2)
    assert len(Q.get_points()) == 3
    dump_element("c1_p1tri", Lagrange(T2, 1), 2, 1, Q.get_points())
    # C2: P3 tet, order 1, degree-6 rule (23 pts) + random points
    Q6 = create_quadrature(T3, 6)
    assert len(Q6.get_points()) == 23
    out["tet_q6_pts"] = Q6.get_points()
    out["tet_q6_wts"] = Q6.get_weights()
    el = Lagrange(T3, 3)
    dump_element("c2_p3tet_q6", el, 3, 1, Q6.get_points())
    out["c2_p3tet_nodes"] = np.array([list(n.get_point_dict().keys())[0] for n in el.dual_basis()])
    rp = np.stack([rand_simplex_points(rng, 3, 23) for _ in range(4)])
    out["c2_p3tet_rand_pts"] = rp
    out["c2_p3tet_rand_tab"] = np.stack([stack(el.tabulate(1, p), 3, 1) for p in rp])
    out["c2_p3tet_o2_tab"] = stack(el.tabulate(2, rp[0]), 3, 2)
    # C2 on physical cells (per-request geometry)
    pv = np.array(T3.get_vertices()) + rng.uniform(-0.2, 0.2, size=(3, 4, 3))
    out["c2_phys_verts"] = pv
    ptabs, ppts, pco = [], [], []
    for v in pv:
        cell = physical_cell(3, v)
        A, b = make_affine_mapping(T3.get_vertices(), v)
        cp = Q6.get_points() @ A.T + b
        elp = Lagrange(cell, 3)
        ptabs.append(stack(elp.tabulate(1, cp), 3, 1))
        ppts.append(cp)
        pco.append(elp.get_coeffs())
    out["c2_phys_pts"] = np.stack(ppts)
    out["c2_phys_tab"] = np.stack(ptabs)
    out["c2_phys_coeffs"] = np.stack(pco)
    # other Lagrange / DG degrees (nodality + tables)
    for sd, cell in ((2, T2), (3, T3)):
        pts = rand_simplex_points(rng, sd, 5)
        for deg in (1, 2, 3, 4):
            dump_element(f"lag_sd{sd}_p{deg}", Lagrange(cell, deg), sd, 2, pts)
            dump_element(f"dg_sd{sd}_p{deg}", DiscontinuousLagrange(cell, deg), sd, 2, pts)
    # C3: N2 + RT2 tet
    for name, cls in (("n2", Nedelec), ("rt2", RaviartThomas)):
        el = cls(T3, 2)
        dump_element(f"c3_{name}tet_q6", el, 3, 1, Q6.get_points())
        out[f"c3_{name}tet_rand_tab"] = np.stack([stack(el.tabulate(1, p), 3, 1) for p in rp[:2]])
    for name, cls in (("n1", Nedelec), ("rt1", RaviartThomas)):
        dump_element(f"c3_{name}tet", cls(T3, 1), 3, 1, rp[0][:5])
        dump_element(f"c3_{name}tri", cls(T2, 1), 2, 1, rand_simplex_points(rng, 2, 5))
    dump_element("c3_n2tri", Nedelec(T2, 2), 2, 1, rand_simplex_points(rng, 2, 5))
    dump_element("c3_rt2tri", RaviartThomas(T2, 2), 2, 1, rand_simplex_points(rng, 2, 5))
    # C4: DG P6 tet, order 2
    el = DiscontinuousLagrange(T3, 6)
    dump_element("c4_dg6tet_q6", el, 3, 2, Q6.get_points())
    out["c4_dg6tet_rand_tab"] = stack(el.tabulate(2, rp[1]), 3, 2)
    Q12 = create_quadrature(T3, 12)
    out["tet_q12_pts"] = Q12.get_points()
    out["c4_dg6tet_q12_tab"] = stack(el.tabulate(2, Q12.get_points()), 3, 2)
    np.savez_compressed(os.path.join(HERE, "elements.npz"), **out)

    # C5: P4 x P4 x P4 hex, order 1
    out = {}
    I = UFCInterval()
    P4 = Lagrange(I, 4)
    out["p4_nodes"] = np.array([list(n.get_point_dict().keys())[0] for n in P4.dual_basis()]).reshape(-1)
    out["p4_coeffs"] = P4.get_coeffs()
    lp = rng.uniform(0, 1, size=(6, 1))
    out["p4_pts"] = lp
    out["p4_tab"] = np.stack([P4.tabulate(2, lp)[(r,)] for r in range(3)])
    hexel = TensorProductElement(TensorProductElement(P4, P4), P4)
    g = np.sort(rng.uniform(0, 1, size=(3, 3)), axis=1)
    grid = np.array([[x, y, z] for x in g[0] for y in g[1] for z in g[2]])
    out["hex_pts"] = grid
    out["hex_tab"] = stack(hexel.tabulate(1, grid), 3, 1)
    rpts = rng.uniform(0, 1, size=(7, 3))
    out["hex_rand_pts"] = rpts
    out["hex_rand_tab"] = stack(hexel.tabulate(1, rpts), 3, 1)
    quad = TensorProductElement(P4, Lagrange(I, 2))
    qp = rng.uniform(0, 1, size=(5, 2))
    out["quad_pts"] = qp
    out["quad_tab"] = stack(quad.tabulate(2, qp), 2, 2)
    # DG1D (Legendre path a1) through the 1-D default expansion set
    es = expansions.ExpansionSet(I)
    out["line_legendre_tab"] = np.stack([es._tabulate(5, lp, 3)[(r,)] for r in range(4)])
    np.savez_compressed(os.path.join(HERE, "tensor_product.npz"), **out)

    # lattices (a16)
    out = {}
    for sd in (1, 2, 3):
        v = np.array(ufc_simplex(sd).get_vertices())
        for n in (1, 2, 3, 6):
            out[f"lattice_sd{sd}_n{n}"] = np.array(make_lattice(v, n))
            out[f"lattice_sd{sd}_n{n}_int1"] = np.array(make_lattice(v, n, 1)).reshape(-1, sd)
    np.savez(os.path.join(HERE, "lattice.npz"), **out)
    print("golden vectors written to", HERE, "FIAT", FIAT.__version__ if hasattr(FIAT, "__version__") else "")


if __name__ == "__main__":
    sys.exit(main())
