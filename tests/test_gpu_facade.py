"""Drop-in facade on the GPU: elements built through the device Vandermonde path
reproduce the reference's nodal coefficients and tables (golden vectors), with the
reference's call signatures and return conventions."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import torch
    assert torch.cuda.is_available()
    import fiat_amd
    return fiat_amd


def stacked(tab, sd, order):
    return np.stack([tab[a] for a in fo.jet_indices(sd, order)])


def relerr(x, ref):
    assert x.shape == ref.shape, (x.shape, ref.shape)
    return float(np.max(np.abs(x - ref)) / max(1.0, np.max(np.abs(ref))))


@pytest.mark.parametrize("sd", [2, 3])
@pytest.mark.parametrize("deg", [1, 2, 3, 4])
def test_lagrange_dg_construction(fa, golden, sd, deg):
    g = golden("elements")
    cell = fa.ufc_simplex(sd)
    for cls, fam in ((fa.Lagrange, "lag"), (fa.DiscontinuousLagrange, "dg")):
        el = cls(cell, deg)
        tag = f"{fam}_sd{sd}_p{deg}"
        assert relerr(el.V, g[f"{tag}_V"]) < 1e-13
        assert relerr(el.get_coeffs(), g[f"{tag}_coeffs"]) < 1e-11
        tab = el.tabulate(2, g[f"{tag}_pts"])
        assert list(tab) == fo.jet_indices(sd, 2)
        assert relerr(stacked(tab, sd, 2), g[f"{tag}_tab"]) < 1e-10
        assert el.space_dimension() == g[f"{tag}_coeffs"].shape[0]
        assert el.value_shape() == ()


def test_test_nodality(fa):
    """test_fiat.py:446-470: dual.to_riesz(P) . coeffs^T = I."""
    for el in (fa.Lagrange(fa.ufc_simplex(2), 3), fa.Lagrange(fa.ufc_simplex(3), 2),
               fa.DiscontinuousLagrange(fa.ufc_simplex(3), 3), fa.DiscontinuousLagrange(fa.ufc_simplex(2), 0),
               fa.Lagrange(fa.ufc_simplex(1), 3), fa.Nedelec(fa.ufc_simplex(3), 1),
               fa.RaviartThomas(fa.ufc_simplex(2), 2)):
        ps = el.get_nodal_basis()
        R = el.get_dual_set().to_riesz(ps)
        co = ps.get_coeffs()
        n = co.shape[0]
        M = R.reshape(n, -1) @ co.reshape(n, -1).T
        assert np.allclose(M, np.eye(n), atol=1e-10)


def test_c2_facade(fa, golden):
    g = golden("elements")
    el = fa.Lagrange(fa.ufc_simplex(3), 3)
    nodes = np.array([list(n.get_point_dict())[0] for n in el.dual_basis()])
    assert relerr(nodes, g["c2_p3tet_nodes"]) < 1e-15
    tab = el.tabulate(1, g["tet_q6_pts"])
    assert relerr(stacked(tab, 3, 1), g["c2_p3tet_q6_tab"]) < 1e-11
    sums = [48.40287093369476, 355.4864717228366, 352.0445676154592, 344.19072083793003]
    assert np.allclose([np.abs(tab[a]).sum() for a in fo.jet_indices(3, 1)], sums, rtol=1e-11)
    # single point drops the trailing axis (test_fiat.py:659-668)
    one = el.tabulate(0, g["tet_q6_pts"][0])[(0, 0, 0)]
    assert one.shape == (20,)
    assert np.allclose(one, tab[(0, 0, 0)][:, 0], atol=1e-13)
    # batched device API agrees with the dict API
    dev = el.tabulate_batch(1, g["c2_p3tet_rand_pts"]).cpu().numpy()
    for o, r in zip(dev, g["c2_p3tet_rand_tab"]):
        assert relerr(o, r) < 1e-11
    assert el.entity_dofs()[0] == {0: [0], 1: [1], 2: [2], 3: [3]}
    assert el.entity_dofs()[3] == {0: []}
    assert el.mapping() == ["affine"] * 20 and el.get_formdegree() == 0


def test_physical_cell_construction(fa, golden):
    g = golden("elements")
    for v, p, ref, cref in zip(g["c2_phys_verts"], g["c2_phys_pts"], g["c2_phys_tab"], g["c2_phys_coeffs"]):
        el = fa.Lagrange(fa.physical_simplex(v), 3)
        assert relerr(el.get_coeffs(), cref) < 1e-10
        assert relerr(stacked(el.tabulate(1, p), 3, 1), ref) < 1e-10


@pytest.mark.parametrize("name,cls,deg,sd", [("n2tet_q6", "Nedelec", 2, 3), ("rt2tet_q6", "RaviartThomas", 2, 3),
                                             ("n1tet", "Nedelec", 1, 3), ("rt1tet", "RaviartThomas", 1, 3),
                                             ("n1tri", "Nedelec", 1, 2), ("rt1tri", "RaviartThomas", 1, 2),
                                             ("n2tri", "Nedelec", 2, 2), ("rt2tri", "RaviartThomas", 2, 2)])
def test_c3_construction(fa, golden, name, cls, deg, sd):
    """N/RT: the SVD spanning basis is not unique, the nodal basis is -- compare
    nodal tables (and coefficients) with the reference's."""
    g = golden("elements")
    el = getattr(fa, cls)(fa.ufc_simplex(sd), deg)
    tag = f"c3_{name}"
    assert el.value_shape() == (sd,)
    assert relerr(el.get_coeffs(), g[f"{tag}_coeffs"]) < 1e-10
    tab = el.tabulate(1, g[f"{tag}_pts"])
    assert relerr(stacked(tab, sd, 1), g[f"{tag}_tab"]) < 1e-10
    assert el.mapping()[0] == ("covariant piola" if cls == "Nedelec" else "contravariant piola")


def test_c4_dg6(fa, golden):
    g = golden("elements")
    el = fa.DiscontinuousLagrange(fa.ufc_simplex(3), 6)
    assert relerr(el.get_coeffs(), g["c4_dg6tet_q6_coeffs"]) < 1e-10
    assert relerr(stacked(el.tabulate(2, g["tet_q6_pts"]), 3, 2), g["c4_dg6tet_q6_tab"]) < 1e-10


def test_interval_and_hex(fa, golden):
    g = golden("tensor_product")
    I = fa.ufc_simplex(1)
    P4 = fa.Lagrange(I, 4)
    assert relerr(P4.get_coeffs(), g["p4_coeffs"]) < 1e-13
    t = P4.tabulate(2, g["p4_pts"])
    assert relerr(np.stack([t[(r,)] for r in range(3)]), g["p4_tab"]) < 1e-12
    hexel = fa.TensorProductElement(fa.TensorProductElement(P4, P4), P4)
    assert hexel.space_dimension() == 125
    tab = hexel.tabulate(1, g["hex_rand_pts"])
    assert relerr(stacked(tab, 3, 1), g["hex_rand_tab"]) < 1e-12
    quad = fa.TensorProductElement(P4, fa.Lagrange(I, 2))
    assert relerr(stacked(quad.tabulate(2, g["quad_pts"]), 2, 2), g["quad_tab"]) < 1e-12
    flat = fa.FlattenedDimensions(hexel)
    assert relerr(stacked(flat.tabulate(1, g["hex_pts"]), 3, 1), g["hex_tab"]) < 1e-12


def test_expansion_set_api(fa, golden):
    g = golden("expansion")
    es = fa.ExpansionSet(fa.ufc_simplex(2))
    tab = es._tabulate(3, g["cpts_sd2_c0"], 2)
    assert relerr(stacked(tab, 2, 2), g["exp_sd2_c0_None_n3_o2"]) < 1e-11
    assert es.tabulate(2, np.array([0.25, 0.5])).shape == g["single_point_tri_n2"].shape
    assert relerr(es.tabulate(2, np.array([0.25, 0.5])), g["single_point_tri_n2"]) < 1e-12
    with pytest.raises(ValueError):
        fa.ExpansionSet(fa.ufc_simplex(2), variant="nope")
    jet = es.tabulate_jet(2, g["cpts_sd2_c0"], 1)
    assert jet[1].shape == (6, 7, 2)


def test_singular_vandermonde_raises(fa):
    """finite_element.py:151-156 -> numpy.linalg.LinAlgError."""
    from fiat_amd import dual_set, finite_element, functional, polynomial_set
    T = fa.ufc_simplex(2)
    nodes = [functional.PointEvaluation(T, (0.2, 0.2))] * 3      # repeated node
    ids = {0: {0: [], 1: [], 2: []}, 1: {0: [], 1: [], 2: []}, 2: {0: [0, 1, 2]}}
    with pytest.raises(np.linalg.LinAlgError):
        finite_element.CiarletElement(polynomial_set.ONPolynomialSet(T, 1), dual_set.DualSet(nodes, T, ids), 1)
    with pytest.raises(ValueError):
        finite_element.CiarletElement(polynomial_set.ONPolynomialSet(T, 2), dual_set.DualSet(nodes, T, ids), 1)


def test_basis_derivatives_scaling_known_answers():
    """The reference's regression test of derivative scaling (test/FIAT/unit/test_fiat.py:76-117, "issue #9"), on the device:
    Lagrange P1 on 26 random intervals [a, b] of length up to 1000, nodal basis tabulated with two derivatives at a, the
    midpoint and b -- values (1, 1/2, 0) / (0, 1/2, 1), first derivatives -+ 1 / (b - a), second derivatives 0; same seed,
    same draws, numpy.isclose as there."""
    import random
    import fiat_amd
    from fiat_amd.reference_element import physical_simplex
    random.seed(42)
    for _ in range(26):
        a = 1000.0 * (random.random() - 0.5)
        b = 1000.0 * (random.random() - 0.5)
        a, b = min(a, b), max(a, b)
        element = fiat_amd.Lagrange(physical_simplex([[a], [b]]), 1)
        points = [(a,), (0.5 * (a + b),), (b,)]
        tab = element.get_nodal_basis().tabulate(points, 2)
        assert np.allclose(tab[(0,)][0], [1.0, 0.5, 0.0]) and np.allclose(tab[(0,)][1], [0.0, 0.5, 1.0])
        D = 1.0 / (b - a)
        for p in range(3):
            assert np.isclose(tab[(1,)][0][p], -D) and np.isclose(tab[(1,)][1][p], +D)
            assert np.isclose(tab[(2,)][0][p], 0.0) and np.isclose(tab[(2,)][1][p], 0.0)


def _outer_rows(*cols):
    """Row a nB + b of a product element = outer product of the factors' columns, last factor fastest."""
    out = cols[0]
    for c in cols[1:]:
        out = np.multiply.outer(out, c).reshape(-1)
    return out


def test_tensor_product_tables_are_products_of_the_factor_tables(fa):
    """The reference's product-rule tests of TensorProductElement.tabulate (test/FIAT/unit/test_tensor_product.py:35-52 DG1 x P2
    on the quadrilateral, :102-123 DG1(triangle) x P2 on the prism, :126-146 (P1 x DG1) x P1 on the hexahedron), at the
    points used there, every multi-index of order <= 1: table[da + db][a nB + b] = A[da][a] B[db][b], numpy.isclose as there."""
    I, S = fa.UFCInterval(), fa.UFCTriangle()
    # 1-D x 1-D
    A, B = fa.DiscontinuousLagrange(I, 1), fa.Lagrange(I, 2)
    elt = fa.TensorProductElement(A, B)
    assert elt.value_shape() == ()
    tab, tA, tB = elt.tabulate(1, [(0.1, 0.2)]), A.tabulate(1, [(0.1,)]), B.tabulate(1, [(0.2,)])
    for da, db in [[(0,), (0,)], [(1,), (0,)], [(0,), (1,)]]:
        assert np.allclose(tab[da + db][:, 0], _outer_rows(tA[da][:, 0], tB[db][:, 0]), rtol=1e-5, atol=1e-8)
    # triangle x interval
    A = fa.DiscontinuousLagrange(S, 1)
    elt = fa.TensorProductElement(A, B)
    assert elt.value_shape() == ()
    tab, tA, tB = elt.tabulate(1, [(0.1, 0.2, 0.3)]), A.tabulate(1, [(0.1, 0.2)]), B.tabulate(1, [(0.3,)])
    for da, db in [[(0, 0), (0,)], [(1, 0), (0,)], [(0, 1), (0,)], [(0, 0), (1,)]]:
        assert np.allclose(tab[da + db][:, 0], _outer_rows(tA[da][:, 0], tB[db][:, 0]), rtol=1e-5, atol=1e-8)
    # (interval x interval) x interval
    P1, D1 = fa.Lagrange(I, 1), fa.DiscontinuousLagrange(I, 1)
    elt = fa.TensorProductElement(fa.TensorProductElement(P1, D1), P1)
    assert elt.value_shape() == ()
    tab = elt.tabulate(1, [(0.1, 0.2, 0.3)])
    tA, tB, tC = P1.tabulate(1, [(0.1,)]), D1.tabulate(1, [(0.2,)]), P1.tabulate(1, [(0.3,)])
    for da, db, dc in [[(0,), (0,), (0,)], [(1,), (0,), (0,)], [(0,), (1,), (0,)], [(0,), (0,), (1,)]]:
        assert np.allclose(tab[da + db + dc][:, 0], _outer_rows(tA[da][:, 0], tB[db][:, 0], tC[dc][:, 0]), rtol=1e-5, atol=1e-8)


def test_flattened_dimensions_equal_the_product_element(fa):
    """test/FIAT/unit/test_tensor_product.py:523-565: FlattenedDimensions of the P1 x P1 quadrilateral and of the
    (P1 x P1) x P1 hexahedron tabulate what the product elements tabulate, keyed by flat multi-indices."""
    I = fa.UFCInterval()
    P1 = fa.Lagrange(I, 1)
    quad = fa.TensorProductElement(P1, P1)
    flat_quad = fa.FlattenedDimensions(quad)
    from fiat_amd.reference_element import UFCHexahedron, UFCQuadrilateral
    assert isinstance(flat_quad.get_reference_element(), UFCQuadrilateral)      # (FIAT/tensor_product.py:373-381)
    assert quad.value_shape() == ()
    t, f = quad.tabulate(1, [(0.1, 0.2)]), flat_quad.tabulate(1, [(0.1, 0.2)])
    for dc in [(0, 0), (1, 0), (0, 1)]:
        assert np.allclose(t[dc], f[dc], rtol=1e-5, atol=1e-8) and t[dc].shape[0] == 4
    hexa = fa.TensorProductElement(quad, P1)
    flat_hex = fa.FlattenedDimensions(fa.TensorProductElement(flat_quad, P1))
    assert isinstance(flat_hex.get_reference_element(), UFCHexahedron)
    assert {d: len(e) for d, e in flat_hex.entity_dofs().items()} == {0: 8, 1: 12, 2: 6, 3: 1}
    t, f = hexa.tabulate(1, [(0.1, 0.2, 0.3)]), flat_hex.tabulate(1, [(0.1, 0.2, 0.3)])
    for dd in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]:
        assert np.allclose(t[dd], f[dd], rtol=1e-5, atol=1e-8) and t[dd].shape[0] == 8


TP_NODAL = ["TensorProductElement(Lagrange(I, 1), Lagrange(I, 1))",
            "TensorProductElement(Lagrange(I, 2), Lagrange(I, 2))",
            "TensorProductElement(TensorProductElement(Lagrange(I, 1), Lagrange(I, 1)), Lagrange(I, 1))",
            "TensorProductElement(TensorProductElement(Lagrange(I, 2), Lagrange(I, 2)), Lagrange(I, 2))",
            "FlattenedDimensions(TensorProductElement(Lagrange(I, 1), Lagrange(I, 1)))",
            "FlattenedDimensions(TensorProductElement(Lagrange(I, 2), Lagrange(I, 2)))",
            "FlattenedDimensions(TensorProductElement(FlattenedDimensions(TensorProductElement(Lagrange(I, 1), Lagrange(I, 1))), Lagrange(I, 1)))",
            "FlattenedDimensions(TensorProductElement(FlattenedDimensions(TensorProductElement(Lagrange(I, 2), Lagrange(I, 2))), Lagrange(I, 2)))"]


@pytest.mark.parametrize("element", TP_NODAL)
def test_nodality_tabulate_of_tensor_product_elements(fa, element):
    """test/FIAT/unit/test_fiat.py:549-581: the eight tensor-product / flattened elements listed there are nodal -- every
    dual node is a point evaluation, and tabulating at node j gives the j-th unit vector (numpy.isclose as there)."""
    from fiat_amd import FlattenedDimensions, Lagrange, TensorProductElement  # noqa: F401  (names of the parametrisation)
    I = fa.UFCInterval()  # noqa: F841
    element = eval(element)
    nodes_coords = []
    for node in element.dual_basis():
        (coords, weights), = node.get_point_dict().items()
        assert weights == [(1.0, ())]
        nodes_coords.append(coords)
    for j, x in enumerate(nodes_coords):
        basis, = element.tabulate(0, (x,)).values()
        for i in range(len(basis)):
            assert np.isclose(basis[i], 1.0 if i == j else 0.0)


def test_constructor_errors_as_in_the_reference(fa):
    """test/FIAT/unit/test_fiat.py:626-650: integral variants with a negative quadrature degree ("integral(-1)") and
    discontinuous elements of degree 1 on a point are ValueErrors."""
    S, P = fa.ufc_simplex(3), fa.ufc_simplex(0)
    for make in (lambda: fa.Nedelec(S, 3, variant="integral(-1)"), lambda: fa.NedelecSecondKind(S, 3, variant="integral(-1)"),
                 lambda: fa.DiscontinuousLagrange(P, 1), lambda: fa.GaussLegendre(P, 1)):
        with pytest.raises(ValueError):
            make()


# The in-scope rows of the parametrisation of the reference's test_nodality (test/FIAT/unit/test_fiat.py:118-445: element
# constructor strings; families outside SURVEY section 8 -- enriched, restricted, bubbles, traces, hierarchical, Bernstein,
# serendipity, histopolation, FDM, discontinuous RT, H(div) / H(curl) wrappers -- left out)
NODAL = [
    'Lagrange(I, 1)', 'Lagrange(I, 2)', 'Lagrange(I, 3)', 'Lagrange(T, 1)', 'Lagrange(T, 2)', 'Lagrange(T, 3)',
    'Lagrange(S, 1)', 'Lagrange(S, 2)', 'Lagrange(S, 3)', 'P0(I)', 'P0(T)', 'P0(S)', 'DiscontinuousLagrange(P, 0)',
    'DiscontinuousLagrange(I, 0)', 'DiscontinuousLagrange(I, 1)', 'DiscontinuousLagrange(I, 2)',
    'DiscontinuousLagrange(T, 0)', 'DiscontinuousLagrange(T, 1)', 'DiscontinuousLagrange(T, 2)',
    'DiscontinuousLagrange(S, 0)', 'DiscontinuousLagrange(S, 1)', 'DiscontinuousLagrange(S, 2)', 'RaviartThomas(I, 1)',
    'RaviartThomas(I, 2)', 'RaviartThomas(I, 3)', 'RaviartThomas(T, 1)', 'RaviartThomas(T, 2)', 'RaviartThomas(T, 3)',
    'RaviartThomas(S, 1)', 'RaviartThomas(S, 2)', 'RaviartThomas(S, 3)', 'RaviartThomas(I, 1, variant="integral")',
    'RaviartThomas(I, 2, variant="integral")', 'RaviartThomas(I, 3, variant="integral")',
    'RaviartThomas(T, 1, variant="integral")', 'RaviartThomas(T, 2, variant="integral")',
    'RaviartThomas(T, 3, variant="integral")', 'RaviartThomas(S, 1, variant="integral")',
    'RaviartThomas(S, 2, variant="integral")', 'RaviartThomas(S, 3, variant="integral")',
    'RaviartThomas(I, 1, variant="integral(1)")', 'RaviartThomas(I, 2, variant="integral(1)")',
    'RaviartThomas(I, 3, variant="integral(1)")', 'RaviartThomas(T, 1, variant="integral(1)")',
    'RaviartThomas(T, 2, variant="integral(1)")', 'RaviartThomas(T, 3, variant="integral(1)")',
    'RaviartThomas(S, 1, variant="integral(1)")', 'RaviartThomas(S, 2, variant="integral(1)")',
    'RaviartThomas(S, 3, variant="integral(1)")', 'RaviartThomas(I, 1, variant="point")',
    'RaviartThomas(I, 2, variant="point")', 'RaviartThomas(I, 3, variant="point")', 'RaviartThomas(T, 1, variant="point")',
    'RaviartThomas(T, 2, variant="point")', 'RaviartThomas(T, 3, variant="point")', 'RaviartThomas(S, 1, variant="point")',
    'RaviartThomas(S, 2, variant="point")', 'RaviartThomas(S, 3, variant="point")', 'BrezziDouglasMarini(T, 1)',
    'BrezziDouglasMarini(T, 2)', 'BrezziDouglasMarini(T, 3)', 'BrezziDouglasMarini(S, 1)', 'BrezziDouglasMarini(S, 2)',
    'BrezziDouglasMarini(S, 3)', 'BrezziDouglasMarini(T, 1, variant="integral")',
    'BrezziDouglasMarini(T, 2, variant="integral")', 'BrezziDouglasMarini(T, 3, variant="integral")',
    'BrezziDouglasMarini(S, 1, variant="integral")', 'BrezziDouglasMarini(S, 2, variant="integral")',
    'BrezziDouglasMarini(S, 3, variant="integral")', 'BrezziDouglasMarini(T, 1, variant="integral(1)")',
    'BrezziDouglasMarini(T, 2, variant="integral(1)")', 'BrezziDouglasMarini(T, 3, variant="integral(1)")',
    'BrezziDouglasMarini(S, 1, variant="integral(1)")', 'BrezziDouglasMarini(S, 2, variant="integral(1)")',
    'BrezziDouglasMarini(S, 3, variant="integral(1)")', 'BrezziDouglasMarini(T, 1, variant="point")',
    'BrezziDouglasMarini(T, 2, variant="point")', 'BrezziDouglasMarini(T, 3, variant="point")',
    'BrezziDouglasMarini(S, 1, variant="point")', 'BrezziDouglasMarini(S, 2, variant="point")',
    'BrezziDouglasMarini(S, 3, variant="point")', 'Nedelec(T, 1)', 'Nedelec(T, 2)', 'Nedelec(T, 3)', 'Nedelec(S, 1)',
    'Nedelec(S, 2)', 'Nedelec(S, 3)', 'Nedelec(T, 1, variant="integral")', 'Nedelec(T, 2, variant="integral")',
    'Nedelec(T, 3, variant="integral")', 'Nedelec(S, 1, variant="integral")', 'Nedelec(S, 2, variant="integral")',
    'Nedelec(S, 3, variant="integral")', 'Nedelec(T, 1, variant="integral(1)")', 'Nedelec(T, 2, variant="integral(1)")',
    'Nedelec(T, 3, variant="integral(1)")', 'Nedelec(S, 1, variant="integral(1)")', 'Nedelec(S, 2, variant="integral(1)")',
    'Nedelec(S, 3, variant="integral(1)")', 'Nedelec(T, 1, variant="point")', 'Nedelec(T, 2, variant="point")',
    'Nedelec(T, 3, variant="point")', 'Nedelec(S, 1, variant="point")', 'Nedelec(S, 2, variant="point")',
    'Nedelec(S, 3, variant="point")', 'NedelecSecondKind(T, 1)', 'NedelecSecondKind(T, 2)', 'NedelecSecondKind(T, 3)',
    'NedelecSecondKind(S, 1)', 'NedelecSecondKind(S, 2)', 'NedelecSecondKind(S, 3)',
    'NedelecSecondKind(T, 1, variant="integral")', 'NedelecSecondKind(T, 2, variant="integral")',
    'NedelecSecondKind(T, 3, variant="integral")', 'NedelecSecondKind(S, 1, variant="integral")',
    'NedelecSecondKind(S, 2, variant="integral")', 'NedelecSecondKind(S, 3, variant="integral")',
    'NedelecSecondKind(T, 1, variant="integral(1)")', 'NedelecSecondKind(T, 2, variant="integral(1)")',
    'NedelecSecondKind(T, 3, variant="integral(1)")', 'NedelecSecondKind(S, 1, variant="integral(1)")',
    'NedelecSecondKind(S, 2, variant="integral(1)")', 'NedelecSecondKind(S, 3, variant="integral(1)")',
    'NedelecSecondKind(T, 1, variant="point")', 'NedelecSecondKind(T, 2, variant="point")',
    'NedelecSecondKind(T, 3, variant="point")', 'NedelecSecondKind(S, 1, variant="point")',
    'NedelecSecondKind(S, 2, variant="point")', 'NedelecSecondKind(S, 3, variant="point")', 'Regge(T, 0)', 'Regge(T, 1)',
    'Regge(T, 2)', 'Regge(S, 0)', 'Regge(S, 1)', 'Regge(S, 2)', "Regge(T, 1, variant='point')",
    "Regge(S, 1, variant='point')", 'HellanHerrmannJohnson(T, 0)', 'HellanHerrmannJohnson(T, 1)',
    'HellanHerrmannJohnson(T, 2)', 'HellanHerrmannJohnson(S, 0)', 'HellanHerrmannJohnson(S, 1)',
    'HellanHerrmannJohnson(S, 2)', "HellanHerrmannJohnson(T, 1, variant='point')",
    "HellanHerrmannJohnson(S, 1, variant='point')", 'GopalakrishnanLedererSchoberlSecondKind(T, 0)',
    'GopalakrishnanLedererSchoberlSecondKind(T, 1)', 'GopalakrishnanLedererSchoberlSecondKind(T, 2)',
    'GopalakrishnanLedererSchoberlSecondKind(S, 0)', 'GopalakrishnanLedererSchoberlSecondKind(S, 1)',
    'GopalakrishnanLedererSchoberlSecondKind(S, 2)', 'GaussLegendre(I, 0)', 'GaussLegendre(I, 1)', 'GaussLegendre(I, 2)',
    'GaussLegendre(T, 0)', 'GaussLegendre(T, 1)', 'GaussLegendre(T, 2)', 'GaussLegendre(S, 0)', 'GaussLegendre(S, 1)',
    'GaussLegendre(S, 2)', 'GaussLobattoLegendre(I, 1)', 'GaussLobattoLegendre(I, 2)', 'GaussLobattoLegendre(I, 3)',
    'GaussLobattoLegendre(T, 1)', 'GaussLobattoLegendre(T, 2)', 'GaussLobattoLegendre(T, 3)', 'GaussLobattoLegendre(S, 1)',
    'GaussLobattoLegendre(S, 2)', 'GaussLobattoLegendre(S, 3)', 'CubicHermite(I)', 'CubicHermite(T)', 'CubicHermite(S)', 'Morley(T)', 'Morley(S)', "Lagrange(T, 1, 'iso')",
    "Lagrange(T, 1, 'alfeld')", "Lagrange(T, 2, 'alfeld')", "DiscontinuousLagrange(T, 1, 'alfeld')",
    ]


@pytest.mark.parametrize("element", NODAL)
def test_nodality_of_every_in_scope_element_of_the_reference_test(fa, element):
    """test/FIAT/unit/test_fiat.py:446-470 on the device: the element's dual basis applied to its nodal basis (Riesz
    representations against expansion coefficients) is the identity, and the nodal basis lives on a cell at least as fine as
    the dual set's.  All 183 in-scope rows, incl. DG on a point and Raviart-Thomas on the interval (polynomials on POINT cells are host arithmetic)."""
    from fiat_amd import (BrezziDouglasMarini, CubicHermite, DiscontinuousLagrange, GaussLegendre, GaussLobattoLegendre,  # noqa: F401
                          GopalakrishnanLedererSchoberlSecondKind, HellanHerrmannJohnson, Lagrange, Morley, Nedelec,
                          NedelecSecondKind, P0, RaviartThomas, Regge)
    P, I, T, S = fa.ufc_simplex(0), fa.UFCInterval(), fa.UFCTriangle(), fa.UFCTetrahedron()  # noqa: F841
    element = eval(element)
    poly_set = element.get_nodal_basis()
    dual_set = element.get_dual_set()
    assert poly_set.get_reference_element() >= dual_set.get_reference_element()
    coeffs_poly = poly_set.get_coeffs()
    coeffs_dual = dual_set.to_riesz(poly_set)
    assert coeffs_poly.shape == coeffs_dual.shape
    n = coeffs_dual.shape[0]
    G = coeffs_dual.reshape(n, -1) @ coeffs_poly.reshape(n, -1).T
    assert np.allclose(G, np.eye(n)), np.abs(G - np.eye(n)).max()
