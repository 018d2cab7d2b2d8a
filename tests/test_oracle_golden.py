"""Pin the CPU oracle (oracle/fiat_oracle.py) against vectors produced by the
unmodified reference (tests/golden/make_golden.py) and against the analytic
known-answer checks of the reference's own suite.  CPU only."""
import math

import numpy as np
import pytest

from oracle import fiat_oracle as fo

TOL = 5e-13


def relerr(x, ref):
    x, ref = np.asarray(x), np.asarray(ref)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(x - ref)) / max(1.0, np.max(np.abs(ref))))


def stacked(tab, sd, order):
    return np.stack([tab[a] for a in fo.jet_indices(sd, order)])


# ---- a9: mis ordering (polynomial_set.py:23-32) -------------------------------
def test_multi_indices_order():
    assert fo.multi_indices(3, 1) == [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
    assert fo.multi_indices(3, 2) == [(2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2)]
    assert fo.multi_indices(2, 2) == [(2, 0), (1, 1), (0, 2)]
    assert fo.multi_indices(1, 3) == [(3,)]


def test_member_index_is_bijection():
    for sd in (2, 3):
        n = 6
        seen = sorted(fo.member_index(i) for i in fo.simplex_lattice(0, n + 1, sd))
        assert seen == list(range(math.comb(n + sd, sd)))


# ---- a1 -------------------------------------------------------------------------
@pytest.mark.parametrize("ab", [(0, 0), (1, 1), (2, 2), (3, 0), (5, 1)])
def test_jacobi(golden, ab):
    g = golden("jacobi")
    a, b = ab
    assert relerr(fo.jacobi_table(a, b, 7, g["jacobi_x"]), g[f"jacobi_{a}_{b}"]) < 1e-14
    assert relerr(fo.jacobi_deriv_table(a, b, 7, g["jacobi_x"]), g[f"jacobi_deriv_{a}_{b}"]) < 1e-14


# ---- a5-a7: raw expansion tables ------------------------------------------------
@pytest.mark.parametrize("sd", [1, 2, 3])
@pytest.mark.parametrize("variant", [None, "bubble", "dual"])
def test_expansion_tables(golden, sd, variant):
    g = golden("expansion")
    checked = 0
    for ci in (0, 1):
        verts = g[f"verts_sd{sd}_c{ci}"]
        pts = g[f"cpts_sd{sd}_c{ci}"]
        for n in (0, 1, 2, 3, 4, 6):
            for order in (0, 1, 2, 3):
                key = f"exp_sd{sd}_c{ci}_{variant}_n{n}_o{order}"
                if key not in g:
                    continue
                tab = fo.expansion_tabulate(verts, n, pts, order, None, variant)
                got = stacked(tab, sd, order)
                assert got.shape == g[key].shape, key
                assert relerr(got, g[key]) < TOL, key
                checked += 1
    assert checked >= 30


@pytest.mark.parametrize("sd", [1, 2, 3])
def test_expansion_physical_cell(golden, sd):
    g = golden("expansion")
    verts, pts = g[f"verts_sd{sd}_phys"], g[f"cpts_sd{sd}_phys"]
    A, b = fo.affine_map(verts, fo.DEFAULT_SIMPLEX[sd])
    assert relerr(A, g[f"affine_A_sd{sd}"]) < 1e-14
    assert relerr(b, g[f"affine_b_sd{sd}"]) < 1e-14
    for variant, scale in ((None, None), ("bubble", 1)):
        for n in (2, 3):
            got = stacked(fo.expansion_tabulate(verts, n, pts, 2, scale, variant), sd, 2)
            assert relerr(got, g[f"exp_sd{sd}_phys_{variant}_n{n}_o2"]) < TOL


def test_single_point_shape(golden):
    g = golden("expansion")
    got = fo.expansion_tabulate(fo.UFC_SIMPLEX[2], 2, np.array([0.25, 0.5]), 0)[(0, 0)]
    assert got.shape == g["single_point_tri_n2"].shape == (6,)
    assert relerr(got, g["single_point_tri_n2"]) < TOL


# ---- analytic anchors restated from the reference's unit tests -------------------
def test_orthonormality_tet():
    """test_polynomial.py:112-120: Dubiner members are L2-orthonormal."""
    from scipy.special import roots_jacobi
    n, sd, m = 4, 3, 6
    rules = [roots_jacobi(m, j, 0) for j in range(sd)]
    pts, wts = [], []
    for i0 in range(m):
        for i1 in range(m):
            for i2 in range(m):
                e = [rules[0][0][i0], rules[1][0][i1], rules[2][0][i2]]
                w = rules[0][1][i0] * rules[1][1][i1] / 2 * rules[2][1][i2] / 4
                x0 = (1 + e[0]) * (1 - e[1]) / 2 * (1 - e[2]) / 2 - 1
                x1 = (1 + e[1]) * (1 - e[2]) / 2 - 1
                pts.append([x0, x1, e[2]])
                wts.append(w)
    pts, wts = np.array(pts), np.array(wts)
    phi = fo.expansion_tabulate(fo.DEFAULT_SIMPLEX[3], n, pts)[(0, 0, 0)]
    gram = (phi * wts) @ phi.T
    assert np.allclose(gram, np.eye(phi.shape[0]), atol=1e-12)


def test_dubiner_closed_form_triangle():
    """test_polynomial.py:34-84 (triangle case): member (p,q) equals the
    collapsed-coordinate Jacobi product."""
    n = 5
    pts = np.array([[-0.5, -0.25], [0.1, -0.7], [-0.9, 0.6], [-1.0 / 3, -1.0 / 3]])
    phi = fo.expansion_tabulate(fo.DEFAULT_SIMPLEX[2], n, pts)[(0, 0)]
    x, y = pts[:, 0], pts[:, 1]
    eta = 2 * (1 + x) / (1 - y) - 1
    for p in range(n + 1):
        for q in range(n + 1 - p):
            Pp = fo.jacobi_table(0, 0, p, eta)[p]
            Pq = fo.jacobi_table(2 * p + 1, 0, q, y)[q]
            expect = Pp * ((1 - y) / 2) ** p * Pq * math.sqrt((p + 0.5) * (p + q + 1))
            assert np.allclose(phi[fo.member_index((p, q))], expect, atol=1e-13)


def test_bubble_dual_duality():
    """test_polynomial.py:123-135: 'bubble' and 'dual' interior members are
    L2-biorthogonal up to a diagonal factor on the interval."""
    from scipy.special import roots_jacobi
    x, w = roots_jacobi(12, 0, 0)
    n = 6
    B = fo.expansion_tabulate(fo.DEFAULT_SIMPLEX[1], n, x[:, None], 0, None, "bubble")[(0,)]
    D = fo.expansion_tabulate(fo.DEFAULT_SIMPLEX[1], n - 2, x[:, None], 0, None, "dual")[(0,)]
    M = (B[2:] * w) @ D.T
    off = M - np.diag(np.diag(M))
    assert np.max(np.abs(off)) < 1e-12


# ---- a8 ------------------------------------------------------------------------
def test_lagrange_line(golden):
    g = golden("lagrange_line")
    D, w = fo.lagrange_dmat(g["nodes"])
    assert relerr(D, g["dmat"]) < 1e-14 and relerr(w, g["wts"]) < 1e-14
    tab = fo.lagrange_line_tabulate(g["nodes"], g["pts"], 2)
    got = np.stack([tab[(r,)] for r in range(3)])
    assert relerr(got, g["tab"]) < 1e-13
    # node hit -> exact Kronecker delta (barycentric_interpolation.py:35-40)
    assert np.array_equal(got[0][:, -1], np.eye(5)[:, 4])


# ---- a16 -----------------------------------------------------------------------
@pytest.mark.parametrize("sd", [1, 2, 3])
def test_lattice(golden, sd):
    g = golden("lattice")
    for n in (1, 2, 3, 6):
        got = np.array(fo.equispaced_lattice(fo.UFC_SIMPLEX[sd], n))
        assert relerr(got, g[f"lattice_sd{sd}_n{n}"]) < 1e-15
        got = np.array(fo.equispaced_lattice(fo.UFC_SIMPLEX[sd], n, 1)).reshape(-1, sd)
        assert relerr(got, g[f"lattice_sd{sd}_n{n}_int1"]) < 1e-15


# ---- a10-a13, a15: nodal elements ----------------------------------------------
@pytest.mark.parametrize("sd", [2, 3])
@pytest.mark.parametrize("deg", [1, 2, 3, 4])
def test_lagrange_and_dg(golden, sd, deg):
    g = golden("elements")
    verts = fo.UFC_SIMPLEX[sd]
    co, V, _ = fo.lagrange_coeffs(verts, deg)
    assert relerr(V, g[f"lag_sd{sd}_p{deg}_V"]) < TOL
    assert relerr(co, g[f"lag_sd{sd}_p{deg}_coeffs"]) < 1e-11
    tab = fo.element_tabulate(verts, deg, co, 2, g[f"lag_sd{sd}_p{deg}_pts"], 1, "bubble")
    assert relerr(stacked(tab, sd, 2), g[f"lag_sd{sd}_p{deg}_tab"]) < 1e-11
    co, V, _ = fo.dg_coeffs(verts, deg)
    assert relerr(V, g[f"dg_sd{sd}_p{deg}_V"]) < TOL
    assert relerr(co, g[f"dg_sd{sd}_p{deg}_coeffs"]) < 1e-11
    tab = fo.element_tabulate(verts, deg, co, 2, g[f"dg_sd{sd}_p{deg}_pts"])
    assert relerr(stacked(tab, sd, 2), g[f"dg_sd{sd}_p{deg}_tab"]) < 1e-11


def test_c1_p1_triangle(golden):
    g = golden("elements")
    co, _, _ = fo.lagrange_coeffs(fo.UFC_SIMPLEX[2], 1)
    tab = fo.element_tabulate(fo.UFC_SIMPLEX[2], 1, co, 1, g["c1_p1tri_pts"], 1, "bubble")
    got = stacked(tab, 2, 1)
    assert relerr(got, g["c1_p1tri_tab"]) < 1e-13
    # sanity anchors recorded in SURVEY.md section 8(c)
    assert np.allclose(got[1], np.array([[-1.0] * 3, [1.0] * 3, [0.0] * 3]), atol=1e-13)
    assert np.allclose(got[2], np.array([[-1.0] * 3, [0.0] * 3, [1.0] * 3]), atol=1e-13)


def test_c2_p3_tet(golden):
    g = golden("elements")
    verts = fo.UFC_SIMPLEX[3]
    co, V, nodes = fo.lagrange_coeffs(verts, 3)
    assert relerr(nodes, g["c2_p3tet_nodes"]) < 1e-15
    assert relerr(co, g["c2_p3tet_q6_coeffs"]) < 1e-12
    got = stacked(fo.element_tabulate(verts, 3, co, 1, g["tet_q6_pts"], 1, "bubble"), 3, 1)
    assert relerr(got, g["c2_p3tet_q6_tab"]) < 1e-12
    sums = [48.40287093369476, 355.4864717228366, 352.0445676154592, 344.19072083793003]
    assert np.allclose(np.abs(got).sum(axis=(1, 2)), sums, rtol=1e-12)
    for p, ref in zip(g["c2_p3tet_rand_pts"], g["c2_p3tet_rand_tab"]):
        got = stacked(fo.element_tabulate(verts, 3, co, 1, p, 1, "bubble"), 3, 1)
        assert relerr(got, ref) < 1e-12
    got = stacked(fo.element_tabulate(verts, 3, co, 2, g["c2_p3tet_rand_pts"][0], 1, "bubble"), 3, 2)
    assert relerr(got, g["c2_p3tet_o2_tab"]) < 1e-11


def test_c2_physical_cells(golden):
    """Lagrange constructed directly on a physical cell (the use exercised by
    test/finat/test_point_evaluation.py:35-70)."""
    g = golden("elements")
    for v, p, ref, cref in zip(g["c2_phys_verts"], g["c2_phys_pts"], g["c2_phys_tab"], g["c2_phys_coeffs"]):
        co, _, _ = fo.lagrange_coeffs(v, 3)
        assert relerr(co, cref) < 1e-11
        got = stacked(fo.element_tabulate(v, 3, co, 1, p, 1, "bubble"), 3, 1)
        assert relerr(got, ref) < 1e-11
    # affine invariance of the coefficients (last bullet)
    ref_co, _, _ = fo.lagrange_coeffs(fo.UFC_SIMPLEX[3], 3)
    assert relerr(g["c2_phys_coeffs"][0], ref_co) < 1e-11


def test_c4_dg6_tet(golden):
    g = golden("elements")
    verts = fo.UFC_SIMPLEX[3]
    co, V, _ = fo.dg_coeffs(verts, 6)
    assert relerr(V, g["c4_dg6tet_q6_V"]) < TOL
    assert relerr(co, g["c4_dg6tet_q6_coeffs"]) < 1e-10
    got = stacked(fo.element_tabulate(verts, 6, co, 2, g["tet_q6_pts"]), 3, 2)
    assert relerr(got, g["c4_dg6tet_q6_tab"]) < 1e-10
    got = stacked(fo.element_tabulate(verts, 6, co, 2, g["tet_q12_pts"]), 3, 2)
    assert relerr(got, g["c4_dg6tet_q12_tab"]) < 1e-10


@pytest.mark.parametrize("name,n", [("n2", 2), ("rt2", 2)])
def test_c3_tabulate_with_reference_coeffs(golden, name, n):
    """Vector-valued contraction (ndof,3,nexp).(nexp,npts) with the reference's
    nodal coefficients as input (polynomial_set.py:68-72)."""
    g = golden("elements")
    co = g[f"c3_{name}tet_q6_coeffs"]
    got = stacked(fo.element_tabulate(fo.UFC_SIMPLEX[3], n, co, 1, g["tet_q6_pts"]), 3, 1)
    assert got.shape == g[f"c3_{name}tet_q6_tab"].shape
    assert relerr(got, g[f"c3_{name}tet_q6_tab"]) < 1e-12


# ---- a14: tensor products ---------------------------------------------------------
def test_c5_hex(golden):
    g = golden("tensor_product")
    nodes = g["p4_nodes"]
    t1 = fo.lagrange_line_tabulate(nodes, g["p4_pts"], 2)
    assert relerr(np.stack([t1[(r,)] for r in range(3)]), g["p4_tab"]) < 1e-12
    got = stacked(fo.hex_lagrange_tabulate(nodes, 1, g["hex_pts"]), 3, 1)
    assert relerr(got, g["hex_tab"]) < 1e-12
    got = stacked(fo.hex_lagrange_tabulate(nodes, 1, g["hex_rand_pts"]), 3, 1)
    assert relerr(got, g["hex_rand_tab"]) < 1e-12


def test_quad_mixed_degree(golden):
    g = golden("tensor_product")
    p = g["quad_pts"]
    ta = fo.lagrange_line_tabulate(g["p4_nodes"], p[:, :1], 2)
    tb = fo.lagrange_line_tabulate(np.array([0.0, 1.0, 0.5]), p[:, 1:], 2)
    got = stacked(fo.tensor_product_tabulate(ta, 1, tb, 1, 2), 2, 2)
    assert relerr(got, g["quad_tab"]) < 1e-12


def test_line_legendre(golden):
    g = golden("tensor_product")
    tab = fo.expansion_tabulate(fo.UFC_SIMPLEX[1], 5, g["p4_pts"], 3)
    assert relerr(np.stack([tab[(r,)] for r in range(4)]), g["line_legendre_tab"]) < 1e-12
