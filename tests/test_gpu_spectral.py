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
