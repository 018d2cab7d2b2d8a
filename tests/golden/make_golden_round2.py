"""Golden vectors of round 2, generated from the *unmodified reference* in the build container:

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tests/golden/make_golden_round2.py

* Jacobi derivative tables of orders 2-4 (FIAT/jacobi.py:85-102);
* create_quadrature: number of points, weights and points of the default and canonical schemes on the UFC interval /
  triangle / tetrahedron over the whole tabulated range and beyond it, on a physical triangle, on a facet
  (FIAT/quadrature_schemes.py:46-106, 324-419);
* derivative orders 3 and 4: raw expansion sets and elements (FIAT/expansions.py:66-137; the relation the reference
  tests in test/FIAT/unit/test_polynomial.py:87-109 and test/FIAT/regression/test_regression.py:283-299);
* general tensor products: triangle x interval prisms, vector-valued factors, facet entities
  (FIAT/tensor_product.py:231-336);
* tabulation on sub-entities, ``entity=(dim, id)`` (FIAT/finite_element.py:181-197, reference_element.py:570-609).
Plain numbers only."""
import os

import numpy as np

from FIAT import (DiscontinuousLagrange, Lagrange, Nedelec, RaviartThomas, TensorProductElement, create_quadrature,
                  expansions, jacobi, ufc_simplex)
from FIAT.polynomial_set import mis
from FIAT.reference_element import UFCSimplex

HERE = os.path.dirname(os.path.abspath(__file__))


def simplex_points(rng, sd, n):
    e = rng.exponential(size=(n, sd + 1))
    return (e / e.sum(axis=1, keepdims=True))[:, 1:].copy()


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for k in range(order + 1) for a in mis(sd, k)])


def main():
    rng = np.random.default_rng(2026)
    out = {}
    # ---- Jacobi ------------------------------------------------------------------------------------
    xs = np.linspace(-1.0, 1.0, 11)[:, None]
    out["jacobi_x"] = xs
    for a, b in [(0, 0), (2, 1), (0.5, 1.5)]:
        for order in (2, 3, 4):
            out[f"jacobi_deriv{order}_{a}_{b}"] = jacobi.eval_jacobi_deriv_batch(a, b, 9, xs, order=order)
    # ---- quadrature --------------------------------------------------------------------------------
    for sd, degrees in ((1, range(0, 12)), (2, range(0, 54)), (3, range(0, 19))):
        cell = ufc_simplex(sd)
        counts = []
        for d in degrees:
            Q = create_quadrature(cell, d)
            counts.append(len(Q.get_weights()))
            out[f"quad_sd{sd}_deg{d}_pts"] = np.asarray(Q.get_points())
            out[f"quad_sd{sd}_deg{d}_wts"] = np.asarray(Q.get_weights())
        out[f"quad_sd{sd}_counts"] = np.array(counts)
        for d in (2, 5):
            Q = create_quadrature(cell, d, "canonical")
            out[f"quadcanon_sd{sd}_deg{d}_pts"], out[f"quadcanon_sd{sd}_deg{d}_wts"] = Q.get_points(), Q.get_weights()
    tri = ufc_simplex(2)
    phys = UFCSimplex(tri.get_shape(), ((0.3, -0.2), (1.7, 0.4), (0.1, 1.9)), tri.get_topology())
    Q = create_quadrature(phys, 7)
    out["quad_phys_tri_verts"], out["quad_phys_tri_pts"], out["quad_phys_tri_wts"] = np.array(phys.get_vertices()), Q.get_points(), Q.get_weights()
    Q = create_quadrature(ufc_simplex(3), 5, entity=(2, 1))
    out["quad_tet_facet1_deg5_pts"], out["quad_tet_facet1_deg5_wts"] = Q.get_points(), Q.get_weights()
    # composite rules on split cells (FIAT/macro.py:381-432; quadrature_schemes.py:71-75)
    from FIAT.macro import AlfeldSplit, IsoSplit, MacroQuadratureRule
    Q = create_quadrature(AlfeldSplit(ufc_simplex(2)), 3)
    out["mq_alfeld_tri_pts"], out["mq_alfeld_tri_wts"] = Q.get_points(), Q.get_weights()
    Q = create_quadrature(IsoSplit(ufc_simplex(3)), 4)
    out["mq_iso_tet_pts"], out["mq_iso_tet_wts"] = Q.get_points(), Q.get_weights()
    Q = MacroQuadratureRule(IsoSplit(ufc_simplex(2)), create_quadrature(ufc_simplex(1), 2), parent_facets=[0, 2])
    out["mq_iso_tri_facets_pts"], out["mq_iso_tri_facets_wts"] = Q.get_points(), Q.get_weights()
    # ---- derivative orders 3, 4 ----------------------------------------------------------------------
    for sd in (1, 2, 3):
        pts = simplex_points(rng, sd, 9)
        out[f"hi_pts_sd{sd}"] = pts
        cell = ufc_simplex(sd)
        for variant in (None, "bubble"):
            es = expansions.ExpansionSet(cell, variant=variant)
            for n in (2, 4):
                tab = es._tabulate(n, pts, order=4)
                out[f"hi_exp_sd{sd}_{variant}_n{n}"] = stack(tab, sd, 4)
    for name, make, sd in (("p3tet", lambda c: Lagrange(c, 3), 3), ("dg4tri", lambda c: DiscontinuousLagrange(c, 4), 2),
                           ("n2tet", lambda c: Nedelec(c, 2), 3), ("p5line", lambda c: Lagrange(c, 5), 1),
                           ("rt3tri", lambda c: RaviartThomas(c, 3), 2), ("dg6tet", lambda c: DiscontinuousLagrange(c, 6), 3)):
        el = make(ufc_simplex(sd))
        for order in (3, 4):
            out[f"hi_{name}_o{order}"] = stack(el.tabulate(order, out[f"hi_pts_sd{sd}"]), sd, order)
    # physical cell: derivatives with respect to the physical coordinates
    tet = ufc_simplex(3)
    verts = np.array(tet.get_vertices()) @ (np.eye(3) + 0.2 * rng.standard_normal((3, 3))).T + rng.standard_normal(3)
    cell = UFCSimplex(tet.get_shape(), tuple(map(tuple, verts)), tet.get_topology())
    e = rng.exponential(size=(9, 4))
    ppts = (e / e.sum(axis=1, keepdims=True)) @ verts
    out["hi_phys_verts"], out["hi_phys_pts"] = verts, ppts
    out["hi_phys_p3tet_o3"] = stack(Lagrange(cell, 3).tabulate(3, ppts), 3, 3)
    # ---- general tensor products ---------------------------------------------------------------------
    I, T = ufc_simplex(1), ufc_simplex(2)
    prism_pts = np.hstack([simplex_points(rng, 2, 7), rng.uniform(0, 1, size=(7, 1))])
    out["tp_prism_pts"] = prism_pts
    cases = {"p2tri_p1": (Lagrange(T, 2), Lagrange(I, 1)), "dg1tri_p2": (DiscontinuousLagrange(T, 1), Lagrange(I, 2)),
             "rt1tri_dg0": (RaviartThomas(T, 1), DiscontinuousLagrange(I, 0)), "n1tri_p1": (Nedelec(T, 1), Lagrange(I, 1)),
             "rt2tri_dg1": (RaviartThomas(T, 2), DiscontinuousLagrange(I, 1))}
    for key, (A, B) in cases.items():
        el = TensorProductElement(A, B)
        for order in (0, 1, 2):
            out[f"tp_{key}_o{order}"] = stack(el.tabulate(order, prism_pts), 3, order)
    # interval x triangle (scalar x vector: the third branch, :318-335)
    pts_it = np.hstack([rng.uniform(0, 1, size=(7, 1)), simplex_points(rng, 2, 7)])
    out["tp_it_pts"] = pts_it
    out["tp_p1_rt1tri_o1"] = stack(TensorProductElement(Lagrange(I, 1), RaviartThomas(T, 1)).tabulate(1, pts_it), 3, 1)
    out["tp_p2_dg1tri_o1"] = stack(TensorProductElement(Lagrange(I, 2), DiscontinuousLagrange(T, 1)).tabulate(1, pts_it), 3, 1)
    # prism facets: entity = ((2, 0), k) bottom/top triangles, ((1, 1), k) side quadrilaterals
    el = TensorProductElement(Lagrange(T, 2), Lagrange(I, 1))
    tri_pts = simplex_points(rng, 2, 5)
    quad_pts = rng.uniform(0, 1, size=(5, 2))
    out["tp_tri_pts"], out["tp_quad_pts"] = tri_pts, quad_pts
    for k in (0, 1):
        out[f"tp_p2tri_p1_ent20_{k}"] = stack(el.tabulate(1, tri_pts, entity=((2, 0), k)), 3, 1)
    for k in (0, 1, 2):
        out[f"tp_p2tri_p1_ent11_{k}"] = stack(el.tabulate(1, quad_pts, entity=((1, 1), k)), 3, 1)
    # facets of a nested (hexahedral) product: entity dimensions are nested tuples, ((1, 1), 0) = bottom / top faces
    P2 = Lagrange(I, 2)
    hexel = TensorProductElement(TensorProductElement(P2, Lagrange(I, 1)), P2)
    out["tp_hex_quad_pts"] = quad_pts
    for dims, count in ((((1, 1), 0), 2), (((1, 0), 1), 2), (((0, 1), 1), 2), (((1, 0), 0), 4)):
        for k in range(count):
            pts_e = quad_pts if sum(dims[0]) + dims[1] == 2 else quad_pts[:, :1]
            key = "".join(str(x) for x in (*dims[0], dims[1]))
            out[f"tp_hex_ent{key}_{k}"] = stack(hexel.tabulate(1, pts_e, entity=(dims, k)), 3, 1)
    out["tp_hex_o3"] = stack(hexel.tabulate(3, prism_pts), 3, 3)
    # the same element on the flattened hexahedron: entities (d, i) numbered by total dimension (tensor_product.py:363-434)
    from FIAT.tensor_product import FlattenedDimensions
    flat = FlattenedDimensions(hexel)
    for k in range(6):
        out[f"tp_flathex_face{k}"] = stack(flat.tabulate(1, quad_pts, entity=(2, k)), 3, 1)
    for k in range(12):
        out[f"tp_flathex_edge{k}"] = stack(flat.tabulate(1, quad_pts[:, :1], entity=(1, k)), 3, 1)
    out["tp_flathex_cell"] = stack(flat.tabulate(1, prism_pts), 3, 1)
    # ---- sub-entity tabulation ------------------------------------------------------------------------
    fpts, epts = simplex_points(rng, 2, 6), rng.uniform(0, 1, size=(6, 1))
    out["ent_facet_pts"], out["ent_edge_pts"] = fpts, epts
    el = Lagrange(ufc_simplex(3), 3)
    for f in range(4):
        out[f"ent_p3tet_facet{f}"] = stack(el.tabulate(1, fpts, entity=(2, f)), 3, 1)
    for e_ in range(6):
        out[f"ent_p3tet_edge{e_}"] = stack(el.tabulate(1, epts, entity=(1, e_)), 3, 1)
    out["ent_p3tet_vertex2"] = stack(el.tabulate(1, np.zeros((1, 0)), entity=(0, 2)), 3, 1)
    el = RaviartThomas(ufc_simplex(2), 2)
    for e_ in range(3):
        out[f"ent_rt2tri_edge{e_}"] = stack(el.tabulate(1, epts, entity=(1, e_)), 2, 1)
    np.savez_compressed(os.path.join(HERE, "round2.npz"), **out)
    print("wrote round2.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "round2.npz")), "bytes")


if __name__ == "__main__":
    main()
