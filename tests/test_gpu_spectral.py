"""Spectral-variant elements on the device: Gauss-Lobatto-Legendre / Gauss-Legendre / Chebyshev Lagrange elements and
the P4-GLL hexahedron against fixtures produced by the reference (tests/golden/make_golden_spectral.py).

Reference: FIAT/gauss_lobatto_legendre.py, FIAT/gauss_legendre.py, FIAT/lagrange.py:75-88 (variant, sort_entities),
FIAT/tensor_product.py:231-292.  1-D elements are pinned mathematically (the node families are closed-form); on simplices
the node placement goes through our restatement of ``recursivenodes`` (parity unpinned w.r.t. that package)."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu

ELEMENTS = {
    "gll_line4": lambda fa: fa.GaussLobattoLegendre(fa.ufc_simplex(1), 4),
    "gll_tri3": lambda fa: fa.GaussLobattoLegendre(fa.ufc_simplex(2), 3),
    "gll_tet3": lambda fa: fa.GaussLobattoLegendre(fa.ufc_simplex(3), 3),
    "cg_spectral_tri4": lambda fa: fa.Lagrange(fa.ufc_simplex(2), 4, "spectral"),
    "gl_line3": lambda fa: fa.GaussLegendre(fa.ufc_simplex(1), 3),
    "gl_tri2": lambda fa: fa.GaussLegendre(fa.ufc_simplex(2), 2),
    "dg_spectral_tet2": lambda fa: fa.DiscontinuousLagrange(fa.ufc_simplex(3), 2, "spectral"),
    "cg_chebyshev_tri3": lambda fa: fa.Lagrange(fa.ufc_simplex(2), 3, "chebyshev"),
}


def rel(x, ref):
    return np.max(np.abs(x - ref)) / max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("name", sorted(ELEMENTS))
def test_spectral_element(golden, name):
    import fiat_amd as fa
    G = golden("spectral")
    e = ELEMENTS[name](fa)
    sd = e.get_reference_element().get_spatial_dimension()
    nodes = np.array([list(ell.get_point_dict().keys())[0] for ell in e.dual_basis()])
    assert np.max(np.abs(nodes - G[f"el/{name}/nodes"])) < 1e-14
    ids = e.entity_dofs()
    flat = [(d, ent, dof) for d in sorted(ids) for ent in sorted(ids[d]) for dof in ids[d][ent]]
    assert np.array_equal(np.array(flat).reshape(-1, 3), G[f"el/{name}/entity_dofs"])
    assert rel(e.get_coeffs(), G[f"el/{name}/coeffs"]) <= 1e-11
    tab = e.tabulate(1, G[f"el/{name}/pts"])
    got = np.stack([tab[a] for a in fo.jet_indices(sd, 1)])
    ref = G[f"el/{name}/tab1"]
    assert rel(got[0], ref[0]) <= 1e-12
    assert all(rel(got[t], ref[t]) <= 1e-10 for t in range(1, got.shape[0]))
    v = e.tabulate(0, nodes)[(0,) * sd]
    assert np.max(np.abs(v - np.eye(len(nodes)))) < 1e-11


def test_gll_hexahedron(golden):
    """The spectral-element hexahedron P4-GLL^3: single call and batched grid form."""
    import fiat_amd as fa
    G = golden("spectral")
    A = fa.GaussLobattoLegendre(fa.ufc_simplex(1), 4)
    hexa = fa.TensorProductElement(fa.TensorProductElement(A, A), A)
    tab = hexa.tabulate(1, G["hex_gll4/pts"])
    got = np.stack([tab[a] for a in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]])
    ref = G["hex_gll4/tab1"]
    assert rel(got[0], ref[0]) <= 1e-12
    assert all(rel(got[t], ref[t]) <= 1e-10 for t in range(1, 4))


# ---- the reference's own tests of the spectral families (test/FIAT/unit/test_gauss_legendre.py, test_gauss_lobatto_legendre.py) ----
# They do not depend on where the interior nodes of the (unpinned) recursivenodes construction sit: the first is a property of
# any unisolvent nodal basis, the second pins the EDGE nodes to the 1-D Gauss / Gauss-Lobatto points.

@pytest.mark.parametrize("family,degrees", [("GaussLegendre", range(0, 8)), ("GaussLobattoLegendre", range(1, 8))])
@pytest.mark.parametrize("dim", (1, 2, 3))
def test_spectral_basis_values_as_in_the_reference_tests(family, degrees, dim):
    """test_gauss_legendre.py:26-43 / test_gauss_lobatto_legendre.py:26-43 with the tabulation on the device: interpolating
    (x_1 + .. + x_d)^k, k <= degree, through the dual nodes and integrating the tabulated basis reproduces the rule's own
    integral of the monomial, rtol 1e-14 as there; symmetric simplex, default rule of degree 2 degree."""
    import fiat_amd
    from fiat_amd import reference_element
    s = reference_element.symmetric_simplex(dim)
    for degree in degrees:
        q = fiat_amd.create_quadrature(s, 2 * degree)
        fe = getattr(fiat_amd, family)(s, degree)
        tab = fe.tabulate(0, q.pts)[(0,) * dim]
        for test_degree in range(degree + 1):
            v = lambda x: sum(x) ** test_degree
            coefs = [n(v) for n in fe.dual.nodes]
            integral = np.dot(coefs, np.dot(tab, q.wts))
            reference = q.integrate(v)
            assert np.allclose(integral, reference, rtol=1e-14), (family, dim, degree, test_degree, integral, reference)


@pytest.mark.parametrize("family", ["GaussLegendre", "GaussLobattoLegendre"])
@pytest.mark.parametrize("dim", (1, 2, 3))
def test_spectral_edge_dofs_as_in_the_reference_tests(family, dim):
    """test_gauss_legendre.py:46-70 / test_gauss_lobatto_legendre.py:46-70, degree 4 on the symmetric simplex: as many dofs as
    P_4 has members, all of them point evaluations, and the edge dofs sit at the images of the 1-D Gauss-Legendre
    (5 points, the edge's own dofs) / Gauss-Lobatto-Legendre points (closure of the edge)."""
    import fiat_amd
    from fiat_amd import expansions, quadrature, reference_element
    degree = 4
    s = reference_element.symmetric_simplex(dim)
    fe = getattr(fiat_amd, family)(s, degree)
    ndof = fe.space_dimension()
    assert ndof == expansions.polynomial_dimension(s, degree)
    points = np.zeros((ndof, dim), "d")
    for i, node in enumerate(fe.dual_basis()):
        points[i, :], = node.get_point_dict().keys()
    line = s if dim == 1 else s.construct_subelement(1)
    if family == "GaussLegendre":     # (interior dofs of the edge at the 5 Gauss points)
        quadrature_points = np.array(quadrature.GaussLegendreQuadratureLineRule(line, degree + 1).pts)
        edge_dofs = fe.entity_dofs()[1]
    else:                             # (closure of the edge: its two vertices and the three interior GLL points)
        quadrature_points = quadrature.GaussLobattoLegendreQuadratureLineRule(line, degree + 1).get_points()
        edge_dofs = fe.entity_closure_dofs()[1]
    for entity in edge_dofs:
        if len(edge_dofs[entity]) > 0:
            transform = s.get_entity_transform(1, entity)
            assert np.allclose(points[edge_dofs[entity]], transform(quadrature_points)), (family, dim, entity)


@pytest.mark.parametrize("family", ["GaussLegendre", "GaussLobattoLegendre"])
@pytest.mark.parametrize("dim,degree", [(1, 64), (2, 16), (3, 16)])
def test_spectral_interpolation_converges_exponentially(family, dim, degree):
    """test_gauss_legendre.py:73-113 / test_gauss_lobatto_legendre.py:73-113: interpolating the Runge function
    1 / (1 + 25 r^2) on the symmetric simplex scaled into the unit ball, degrees 1, 2, 4, .. -- the maximum error at a GL
    lattice of 2 degree + 1 points per edge stays below 2 C^-k, C = sqrt(1/25) + sqrt(1 + 1/25).  Elements built and tabulated
    on the device, at the reference's sizes: 65 nodes on the interval (the many-node 1-D kernel), 153 dofs on the triangle, 969 on
    the tetrahedron (one 124 KB column tile of the generic kernel per wave)."""
    import fiat_amd
    from fiat_amd import reference_element
    s = reference_element.symmetric_simplex(dim)
    radius = max(np.linalg.norm(s.get_vertices(), axis=-1))
    s = reference_element.SymmetricSimplex(s.get_shape(), np.array(s.get_vertices()) / radius, s.get_topology())
    A = 25
    f = lambda x: 1 / (1 + A * np.linalg.norm(x, axis=-1) ** 2)
    points = np.array(reference_element.make_lattice(s.get_vertices(), 2 * degree + 1, variant="gl"))
    f_at_pts = f(points)
    k, errors, degrees = 1, [], []
    while k <= degree:
        fe = getattr(fiat_amd, family)(s, k)
        coefficients = np.array([v(f) for v in fe.dual_basis()])
        tab = fe.tabulate(0, points)[(0,) * dim]
        errors.append(max(abs(f_at_pts - np.dot(coefficients, tab))))
        degrees.append(k)
        k *= 2
    C = np.sqrt(1 / A) + np.sqrt(1 + 1 / A)
    assert all(np.array(errors) < 2.0 * C ** -np.array(degrees)), (errors, degrees)


@pytest.mark.parametrize("family", ["GaussLegendre", "GaussLobattoLegendre"])
@pytest.mark.parametrize("degree", [4, 8, 12, 16])
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_spectral_conditioning_as_in_the_reference_tests(family, dim, degree):
    """test_gauss_legendre.py:116-141 / test_gauss_lobatto_legendre.py:116-141: condition numbers of the mass and the
    stiffness matrix of the spectral bases grow at most like (dim + 1)^degree / (dim + 2)^degree; basis and gradients
    tabulated on the device at the default rule of degree 2 degree."""
    import fiat_amd
    from fiat_amd import reference_element
    s = reference_element.symmetric_simplex(dim)
    rule = fiat_amd.create_quadrature(s, 2 * degree)
    points, weights = rule.get_points(), rule.get_weights()
    fe = getattr(fiat_amd, family)(s, degree)
    phi = fe.tabulate(1, points)
    v = phi[(0,) * dim]
    grads = [phi[alpha] for alpha in phi if sum(alpha) == 1]
    M = np.dot(v, weights[:, None] * v.T)
    K = sum(np.dot(dv, weights[:, None] * dv.T) for dv in grads)

    def cond(A):
        a = np.linalg.eigvalsh(A)
        a = a[abs(a) > 1E-12]
        return max(a) / min(a)

    assert cond(M) ** (1 / degree) < dim + 1
    assert cond(K) ** (1 / degree) < dim + 2
