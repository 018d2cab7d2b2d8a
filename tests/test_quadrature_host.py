"""create_quadrature against the reference's own output (tests/golden/round2.npz, make_golden_round2.py): default
scheme = classical / Xiao-Gimbutas tables on triangles (degree <= 50) and tetrahedra (<= 15), collapsed Gauss-Jacobi
beyond and for "canonical"; exactness as in test/FIAT/unit/test_quadrature.py:110-125.  Host logic only."""
import math

import numpy as np
import pytest

import fiat_amd
from fiat_amd import quadrature
from fiat_amd.check_format_variant import parse_quadrature_scheme


@pytest.mark.parametrize("sd,degrees", [(1, range(0, 12)), (2, range(0, 54)), (3, range(0, 19))])
def test_default_scheme_equals_reference(golden, sd, degrees):
    g = golden("round2")
    cell = fiat_amd.ufc_simplex(sd)
    counts = []
    for d in degrees:
        Q = fiat_amd.create_quadrature(cell, d)
        counts.append(len(Q.get_weights()))
        assert Q.get_points().shape == g[f"quad_sd{sd}_deg{d}_pts"].shape, d
        assert np.abs(Q.get_points() - g[f"quad_sd{sd}_deg{d}_pts"]).max() <= 1e-14, d
        assert np.abs(Q.get_weights() - g[f"quad_sd{sd}_deg{d}_wts"]).max() <= 1e-14, d
    assert counts == list(g[f"quad_sd{sd}_counts"])
    for d in (2, 5):
        Q = fiat_amd.create_quadrature(cell, d, "canonical")
        assert np.abs(Q.get_points() - g[f"quadcanon_sd{sd}_deg{d}_pts"]).max() <= 1e-14
        assert np.abs(Q.get_weights() - g[f"quadcanon_sd{sd}_deg{d}_wts"]).max() <= 1e-14


def test_headline_rule_and_other_cells(golden):
    g = golden("round2")
    Q = fiat_amd.create_quadrature(fiat_amd.ufc_simplex(3), 6)
    assert len(Q.get_weights()) == 23                     # the rule BASELINE configs[1] is defined on
    assert np.array_equal(Q.get_points(), golden("elements")["tet_q6_pts"])
    cell = fiat_amd.physical_simplex(g["quad_phys_tri_verts"])
    Q = fiat_amd.create_quadrature(cell, 7)
    assert np.abs(Q.get_points() - g["quad_phys_tri_pts"]).max() <= 1e-14
    assert np.abs(Q.get_weights() - g["quad_phys_tri_wts"]).max() <= 1e-14
    Q = fiat_amd.create_quadrature(fiat_amd.ufc_simplex(3), 5, entity=(2, 1))
    assert np.abs(Q.get_points() - g["quad_tet_facet1_deg5_pts"]).max() <= 1e-14
    assert np.abs(Q.get_weights() - g["quad_tet_facet1_deg5_wts"]).max() <= 1e-14
    assert quadrature.tabulated_rule(2, 51) is None and quadrature.tabulated_rule(3, 16) is None
    assert quadrature.tabulated_rule(1, 3) is None


@pytest.mark.parametrize("sd,maxdeg", [(2, 50), (3, 15)])
def test_exactness_of_the_tabulated_rules(sd, maxdeg):
    """Every monomial x^a y^b (z^c) of total degree <= d integrates to a! b! c! / (a + b + c + sd)!."""
    cell = fiat_amd.ufc_simplex(sd)
    for d in list(range(0, maxdeg + 1, 3)) + [maxdeg]:
        Q = fiat_amd.create_quadrature(cell, d)
        x, w = Q.get_points(), Q.get_weights()
        for alpha in fiat_amd.mis(sd, d):
            exact = math.prod(math.factorial(a) for a in alpha) / math.factorial(sum(alpha) + sd)
            got = float(np.sum(w * np.prod(x ** np.array(alpha), axis=1)))
            assert abs(got - exact) <= 2e-13 * max(exact, 1e-3), (d, alpha)


def test_scheme_strings():
    tri = fiat_amd.ufc_simplex(2)
    with pytest.raises(ValueError):
        fiat_amd.create_quadrature(tri, 2, "no-such-scheme")
    with pytest.raises(ValueError):
        fiat_amd.create_quadrature(tri, -1)
    with pytest.raises(NotImplementedError):
        fiat_amd.create_quadrature(tri, 2, "KMV")
    assert len(parse_quadrature_scheme(tri, 4).get_weights()) == 6
    assert len(parse_quadrature_scheme(tri, 4, "canonical").get_weights()) == 9
    Q = parse_quadrature_scheme(tri, 4, "default,alfeld")      # composite rule on the Alfeld split: 3 sub-triangles
    assert len(Q.get_weights()) == 18 and abs(Q.get_weights().sum() - 0.5) < 1e-15
    with pytest.raises(NotImplementedError):
        parse_quadrature_scheme(tri, 4, "KMV(2)")


def test_composite_rules_on_split_cells(golden):
    """create_quadrature on a macro cell and MacroQuadratureRule on the children of parent facets
    (FIAT/macro.py:381-432, quadrature_schemes.py:71-75): the same point sets (as sets: the merge order of coincident
    points is an implementation detail) and weights as the reference."""
    g = golden("round2")

    def same(Q, key):
        pts, wts = np.asarray(Q.get_points()), np.asarray(Q.get_weights())
        rp, rw = g[key + "_pts"], g[key + "_wts"]
        assert pts.shape == rp.shape and abs(wts.sum() - rw.sum()) < 1e-14
        order, rorder = np.lexsort(np.round(pts, 12).T), np.lexsort(np.round(rp, 12).T)
        assert np.abs(pts[order] - rp[rorder]).max() < 1e-13 and np.abs(wts[order] - rw[rorder]).max() < 1e-14

    same(fiat_amd.create_quadrature(fiat_amd.AlfeldSplit(fiat_amd.ufc_simplex(2)), 3), "mq_alfeld_tri")
    same(fiat_amd.create_quadrature(fiat_amd.IsoSplit(fiat_amd.ufc_simplex(3)), 4), "mq_iso_tet")
    from fiat_amd.macro import MacroQuadratureRule
    same(MacroQuadratureRule(fiat_amd.IsoSplit(fiat_amd.ufc_simplex(2)), fiat_amd.create_quadrature(fiat_amd.ufc_simplex(1), 2),
                             parent_facets=[0, 2]), "mq_iso_tri_facets")


def test_line_rules_and_argument_errors_as_in_the_reference_tests():
    """test/FIAT/unit/test_quadrature.py:104-108 (points / weights mismatch is a ValueError), :166-168 (negative degree is a
    ValueError), :187-220: the m-point Gauss-Lobatto-Legendre / Gauss-Radau / Gauss-Legendre rules on the UFC interval integrate
    x^d exactly for d < 2m - 2 / d < 2m, m = 2..9 (numpy.round(error, 14) == 0 as there)."""
    import fiat_amd
    from fiat_amd import quadrature
    interval = fiat_amd.ufc_simplex(1)
    with pytest.raises(ValueError):
        quadrature.QuadratureRule(interval, [[0.5, 0.5]], [0.5, 0.5, 0.5])
    for cell in (interval, fiat_amd.ufc_simplex(2), fiat_amd.ufc_simplex(3)):
        with pytest.raises(ValueError):
            fiat_amd.create_quadrature(cell, -1)
    for points in range(2, 10):
        gll = quadrature.GaussLobattoLegendreQuadratureLineRule(interval, points)
        for degree in range(2 * points - 2):
            assert np.round(gll.integrate(lambda x: x[0] ** degree) - 1. / (degree + 1), 14) == 0.
        gl = quadrature.GaussLegendreQuadratureLineRule(interval, points)
        for degree in range(2 * points):
            assert np.round(gl.integrate(lambda x: x[0] ** degree) - 1. / (degree + 1), 14) == 0.
        for right in (True, False):      # (:199-208: Gauss-Radau, exact for d < 2m - 1, the fixed end point among the nodes)
            radau = quadrature.RadauQuadratureLineRule(interval, points, right=right)
            assert np.isclose(radau.get_points().ravel()[-1 if right else 0], 1.0 if right else 0.0)
            for degree in range(2 * points - 1):
                assert np.round(radau.integrate(lambda x: x[0] ** degree) - 1. / (degree + 1), 14) == 0.


def test_quadrature_on_product_cells_as_in_the_reference_tests():
    """test/FIAT/unit/test_quadrature.py:128-141, :171-184: create_quadrature on interval x interval and triangle x interval
    with a degree per factor integrates x^a y^b / (x + y)^a z^b exactly for a < 5, b < 4; negative degrees are ValueErrors;
    the rule of a product is the product of the factors' rules (point counts)."""
    import fiat_amd
    from fiat_amd import reference_element as re
    I, T = re.ufc_simplex(1), re.ufc_simplex(2)
    extr_interval, extr_triangle = re.TensorProductCell(I, I), re.TensorProductCell(T, I)
    for basedeg in range(5):
        for extrdeg in range(4):
            q = fiat_amd.create_quadrature(extr_interval, (basedeg, extrdeg))
            assert np.allclose(q.integrate(lambda x: x[0] ** basedeg * x[1] ** extrdeg), 1 / (basedeg + 1) * 1 / (extrdeg + 1))
            q = fiat_amd.create_quadrature(extr_triangle, (basedeg, extrdeg))
            assert np.allclose(q.integrate(lambda x: (x[0] + x[1]) ** basedeg * x[2] ** extrdeg), 1 / (basedeg + 2) * 1 / (extrdeg + 1))
    for cell in (extr_interval, extr_triangle):
        with pytest.raises(ValueError):
            fiat_amd.create_quadrature(cell, (-1, -1))
    qa, qb = fiat_amd.create_quadrature(T, 4), fiat_amd.create_quadrature(I, 4)
    q = fiat_amd.create_quadrature(extr_triangle, (4, 4))
    assert len(q.get_points()) == len(qa.get_points()) * len(qb.get_points())


def test_quadrature_on_hypercubes_as_in_the_reference_tests():
    """test/FIAT/unit/test_quadrature.py:143-163: create_quadrature on the UFC quadrilateral / hexahedron integrates
    (x + y [+ z])^d exactly, d < 8, and on quadrilateral x interval (x + y)^a z^b for a < 5, b < 4."""
    import fiat_amd
    from fiat_amd import reference_element as re
    Q, H = re.UFCQuadrilateral(), re.UFCHexahedron()
    for degree in range(8):
        q = fiat_amd.create_quadrature(Q, degree)
        assert np.allclose(q.integrate(lambda x: sum(x) ** degree), (2 ** (degree + 2) - 2) / ((degree + 1) * (degree + 2)))
        q = fiat_amd.create_quadrature(H, degree)
        assert np.allclose(q.integrate(lambda x: sum(x) ** degree),
                           -3 * (2 ** (degree + 3) - 3 ** (degree + 2) - 1) / ((degree + 1) * (degree + 2) * (degree + 3)))
    extr = re.TensorProductCell(Q, re.ufc_simplex(1))
    for basedeg in range(5):
        for extrdeg in range(4):
            q = fiat_amd.create_quadrature(extr, (basedeg, extrdeg))
            assert np.allclose(q.integrate(lambda x: (x[0] + x[1]) ** basedeg * x[2] ** extrdeg),
                               (2 ** (basedeg + 2) - 2) / ((basedeg + 1) * (basedeg + 2)) * 1 / (extrdeg + 1))
