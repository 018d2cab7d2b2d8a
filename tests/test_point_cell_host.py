"""Elements whose definition needs polynomials on a POINT cell (FIAT/expansions.py:638-649): DG on a point, and the
integral variants of Raviart-Thomas on the interval (its facets are points).  Host logic with the oracle standing in for the
device (tests/host_backend.py); the same rows run on the device in tests/test_gpu_facade.py (test_nodality...)."""
import numpy as np
import pytest

from host_backend import oracle_backend  # noqa: F401


@pytest.mark.parametrize("make", ["DiscontinuousLagrange(P, 0)", "RaviartThomas(I, 1)", "RaviartThomas(I, 2)", "RaviartThomas(I, 3)",
                                  'RaviartThomas(I, 2, variant="integral(1)")'])
def test_nodality_on_and_over_point_cells(oracle_backend, make):  # noqa: F811
    import fiat_amd
    from fiat_amd import DiscontinuousLagrange, RaviartThomas  # noqa: F401
    P, I = fiat_amd.ufc_simplex(0), fiat_amd.ufc_simplex(1)  # noqa: F841
    element = eval(make)
    poly_set = element.get_nodal_basis()
    coeffs_poly = poly_set.get_coeffs()
    coeffs_dual = element.get_dual_set().to_riesz(poly_set)
    n = coeffs_dual.shape[0]
    assert coeffs_poly.shape == coeffs_dual.shape
    assert np.allclose(coeffs_dual.reshape(n, -1) @ coeffs_poly.reshape(n, -1).T, np.eye(n))
