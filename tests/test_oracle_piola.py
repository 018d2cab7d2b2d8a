"""CPU: the push-forward formula, evaluated with the oracle, against the reference's elements
constructed directly on physical cells (tests/golden/piola.npz)."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

EXACT = [("n1", 1, "cov", 2), ("n1", 1, "cov", 3), ("n2", 2, "cov", 3), ("rt1", 1, "con", 2), ("rt1", 1, "con", 3)]


@pytest.mark.parametrize("name,n,kind,sd", EXACT)
def test_piola_image_of_reference_basis_is_the_physical_basis(golden, name, n, kind, sd):
    g = golden("piola")
    ref_cell = fo.UFC_SIMPLEX[sd]
    co = g[f"{name}_sd{sd}_refcoeffs"]
    for v, p, gold in zip(g[f"verts_sd{sd}"], g[f"pts_sd{sd}"], g[f"{name}_sd{sd}_tab"]):
        J = (v[1:] - v[0]).T @ np.linalg.inv((ref_cell[1:] - ref_cell[0]).T)
        M = np.linalg.inv(J).T if kind == "cov" else J / np.linalg.det(J)
        tab = fo.element_tabulate(v, n, co, 1, p)
        raw = np.stack([tab[a] for a in fo.jet_indices(sd, 1)])
        got = np.einsum("ce,tdep->tdcp", M, raw)
        assert np.abs(got - gold).max() <= 5e-12 * max(1.0, np.abs(gold).max())


@pytest.mark.parametrize("sd", [2, 3])
def test_affine_pullback(golden, sd):
    g = golden("piola")
    co = g[f"p2_sd{sd}_refcoeffs"]
    for v, p, gold in zip(g[f"verts_sd{sd}"], g[f"pts_sd{sd}"], g[f"p2_sd{sd}_tab"]):
        tab = fo.element_tabulate(v, 2, co, 1, p, 1, "bubble")
        raw = np.stack([tab[a] for a in fo.jet_indices(sd, 1)])
        assert np.abs(raw - gold).max() <= 1e-12 * max(1.0, np.abs(gold).max())
