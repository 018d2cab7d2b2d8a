"""Point production on the device: Gauss-Jacobi / collapsed simplex rules
through the C ABI against SciPy's roots_jacobi, the host facade's rule and exact integrals."""
import math

import numpy as np
import pytest
from scipy.special import roots_jacobi

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from fiat_amd import runtime
    runtime.Context.get()
    return runtime


@pytest.mark.parametrize("m", [1, 2, 3, 5, 8, 13, 24, 40])
def test_gauss_legendre_line(rt, m):
    pts, wts = rt.collapsed_quadrature(1, m)
    x, w = roots_jacobi(m, 0, 0)
    assert np.abs(pts.cpu().numpy()[:, 0] - 0.5 * (x + 1.0)).max() <= 2e-15
    assert np.abs(wts.cpu().numpy() - 0.5 * w).max() <= 5e-15


@pytest.mark.parametrize("sd", [2, 3])
@pytest.mark.parametrize("m", [1, 2, 4, 7, 12])
def test_collapsed_rule_matches_the_host_rule(rt, sd, m):
    import fiat_amd
    from fiat_amd import quadrature
    ref_el = fiat_amd.ufc_simplex(sd)
    Q = quadrature.CollapsedQuadratureSimplexRule(ref_el, m)
    pts, wts = rt.collapsed_quadrature(sd, m)
    assert np.abs(pts.cpu().numpy() - Q.get_points()).max() <= 1e-14
    assert np.abs(wts.cpu().numpy() - Q.get_weights()).max() <= 1e-14
    assert abs(float(wts.sum()) - 1.0 / math.factorial(sd)) <= 1e-14


def test_rule_on_a_physical_cell_integrates_monomials(rt):
    rng = np.random.default_rng(4)
    verts = fo.UFC_SIMPLEX[3] @ (np.eye(3) + 0.2 * rng.standard_normal((3, 3))).T + rng.standard_normal(3)
    m = 5                                   # exact to degree 9
    pts, wts = rt.collapsed_quadrature(3, m, verts=verts)
    p, w = pts.cpu().numpy(), wts.cpu().numpy()
    # exact integral of a polynomial over the physical tet = vol * (reference integral of the pulled-back polynomial):
    # compare with a much finer rule from the host facade
    import fiat_amd
    from fiat_amd import quadrature, reference_element
    cell = reference_element.UFCSimplex(fiat_amd.ufc_simplex(3).get_shape(), tuple(map(tuple, verts)),
                                        fiat_amd.ufc_simplex(3).get_topology())
    Q = quadrature.CollapsedQuadratureSimplexRule(cell, 9)
    for (i, j, k) in [(0, 0, 0), (1, 0, 0), (2, 1, 0), (3, 3, 3), (0, 4, 5), (9, 0, 0)]:
        f = lambda x: x[:, 0] ** i * x[:, 1] ** j * x[:, 2] ** k
        ref = float(np.dot(Q.get_weights(), f(Q.get_points())))
        got = float(np.dot(w, f(p)))
        assert abs(got - ref) <= 1e-12 * max(1.0, abs(ref)), ((i, j, k), got, ref)


def test_device_rule_feeds_the_shared_point_tabulation(rt, golden):
    """rule (device) -> tabulate_cells (device): mass matrix of P2 on a physical triangle equals
    |K| / 2 * the reference mass matrix."""
    import fiat_amd
    el = fiat_amd.Lagrange(fiat_amd.ufc_simplex(2), 2)
    pts, wts = rt.collapsed_quadrature(2, 4)
    verts = np.array([[[0.3, -0.2], [1.7, 0.1], [0.2, 1.4]]])
    tab = el.tabulate_cells(0, pts, verts)[0, 0]          # (ndof, npts), values are affine invariant
    area = 0.5 * abs(np.linalg.det(verts[0][1:] - verts[0][0]))
    M = (tab * wts) @ tab.T * (area / 0.5)
    Mref = np.array([[6, -1, -1, 0, -4, 0], [-1, 6, -1, -4, 0, 0], [-1, -1, 6, 0, 0, -4],
                     [0, -4, 0, 32, 16, 16], [-4, 0, 0, 16, 32, 16], [0, 0, -4, 16, 16, 32]]) * area / 180.0
    got = M.cpu().numpy()
    # FIAT orders the edge dofs of P2 as (edge 0: v1-v2, edge 1: v0-v2, edge 2: v0-v1)
    assert np.abs(np.sort(np.diag(got)) - np.sort(np.diag(Mref))).max() <= 1e-13
    assert abs(got.sum() - area) <= 1e-13


def test_bad_arguments(rt):
    with pytest.raises(ValueError):
        rt.collapsed_quadrature(3, 0)
    with pytest.raises(ValueError):
        rt.collapsed_quadrature(4, 2)
