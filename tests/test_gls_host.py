"""CPU check of the GLS element DEFINITION and the trace-free tensor space (fiat_amd/gopalakrishnan_lederer_schoberl.py,
polynomial_set.TracelessTensorPolynomialSet) against the reference's numbers (tests/golden/round3.npz:
FIAT/polynomial_set.py:252-282, FIAT/gopalakrishnan_lederer_schoberl.py:9-71), with the oracle standing in for the device
arithmetic (tests/host_backend.py).  The same cases run on the HIP path in tests/test_gpu_round3.py."""
import json

import numpy as np
import pytest

from host_backend import oracle_backend  # noqa: F401  (fixture)

GLS = [(2, 0), (2, 1), (2, 2), (3, 0), (3, 1)]


def rel(x, ref):
    return np.abs(x - ref).max() / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("sd,k", GLS)
def test_traceless_space_equals_reference(golden, sd, k):
    import fiat_amd
    P = fiat_amd.TracelessTensorPolynomialSet(fiat_amd.ufc_simplex(sd), k)
    want = golden("round3")[f"gls_sd{sd}_k{k}_space"]
    assert P.get_coeffs().shape == want.shape and np.array_equal(P.get_coeffs(), want)
    assert np.abs(np.trace(P.get_coeffs(), axis1=1, axis2=2)).max() == 0.0


@pytest.mark.parametrize("sd,k", GLS)
def test_gls_definition(golden, oracle_backend, sd, k):  # noqa: F811
    import fiat_amd
    g = golden("round3")
    key = f"gls_sd{sd}_k{k}"
    el = fiat_amd.GopalakrishnanLedererSchoberlSecondKind(fiat_amd.ufc_simplex(sd), k)
    assert el.get_coeffs().shape == g[key + "_coeffs"].shape
    assert rel(el.get_coeffs(), g[key + "_coeffs"]) <= 1e-11
    assert el.mapping()[0] == str(g[key + "_mapping"]) == "covariant contravariant piola"
    want = json.loads(str(g[key + "_entity_dofs"]))
    assert {str(d): {str(i): list(v) for i, v in ents.items()} for d, ents in el.entity_dofs().items()} == want
    assert el.value_shape() == (sd, sd) and el.get_formdegree() == (1, sd - 1)
    tab = el.tabulate(1, g[key + "_pts"])
    for t, a in enumerate([a for j in range(2) for a in fiat_amd.mis(sd, j)]):
        assert rel(tab[a], g[key + "_tab"][t]) <= 1e-10, a
    # trace-free basis functions
    assert np.abs(np.trace(tab[(0,) * sd], axis1=1, axis2=2)).max() <= 1e-12
