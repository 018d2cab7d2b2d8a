"""CPU check of the element DEFINITIONS (fiat_amd/dof_layout.py and the family tables over it): which
functionals sit on which entity, in which order, and which polynomial space -- with the oracle standing in
for the device arithmetic (tests/host_backend.py).  Expected numbers: the reference's own nodal
coefficients, entity dofs and tables (tests/golden/families.npz, elements.npz).  The same cases run on the
HIP path in test_gpu_families.py / test_gpu_facade.py."""
import json

import numpy as np
import pytest

from host_backend import oracle_backend  # noqa: F401  (fixture)
from test_gpu_families import CASES, MORE

ALL = [(n, (lambda fa, c, k, cls=cls: getattr(fa, cls)(c, k)), sd, k) for n, cls, sd, k in CASES] + list(MORE)


def rel(x, ref):
    return np.abs(x - ref).max() / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,make,sd,k", ALL, ids=[f"{m[0]}{m[3]}_sd{m[2]}" for m in ALL])
def test_family_definitions(golden, oracle_backend, name, make, sd, k):  # noqa: F811
    import fiat_amd
    g = golden("families")
    key = f"{name}{k}_sd{sd}"
    el = make(fiat_amd, fiat_amd.ufc_simplex(sd), k)
    assert rel(el.get_coeffs(), g[key + "_coeffs"]) <= 1e-11
    assert el.mapping()[0] == str(g[key + "_mapping"])
    want = json.loads(str(g[key + "_entity_dofs"]))
    assert {str(d): {str(i): list(v) for i, v in ents.items()} for d, ents in el.entity_dofs().items()} == want
    tab = el.tabulate(1, g[f"pts_sd{sd}"])
    for t, a in enumerate([a for j in range(2) for a in fiat_amd.mis(sd, j)]):
        assert rel(tab[a], g[key + "_tab"][t]) <= 1e-10, a


BASE = [("c3_n2tet_q6", "Nedelec", 3, 2), ("c3_rt2tet_q6", "RaviartThomas", 3, 2), ("c3_n1tet", "Nedelec", 3, 1),
        ("c3_n1tri", "Nedelec", 2, 1), ("c3_rt1tet", "RaviartThomas", 3, 1), ("c3_rt1tri", "RaviartThomas", 2, 1),
        ("c3_n2tri", "Nedelec", 2, 2), ("c3_rt2tri", "RaviartThomas", 2, 2)] + \
       [(f"{fam}_sd{sd}_p{p}", cls, sd, p) for fam, cls in (("lag", "Lagrange"), ("dg", "DiscontinuousLagrange"))
        for sd in (2, 3) for p in (1, 2, 3, 4)]


@pytest.mark.parametrize("key,cls,sd,k", BASE, ids=[b[0] for b in BASE])
def test_baseline_families(golden, oracle_backend, key, cls, sd, k):  # noqa: F811
    """Lagrange / DG / Nedelec / Raviart-Thomas of the BASELINE configs: nodal coefficients and the Vandermonde
    matrix V equal the reference's (any basis of the N / RT spaces gives the same nodal basis; V itself depends
    on the spanning basis, so it is compared for the point-evaluation families only)."""
    import fiat_amd
    g = golden("elements")
    el = getattr(fiat_amd, cls)(fiat_amd.ufc_simplex(sd), k)
    assert rel(el.get_coeffs(), g[key + "_coeffs"]) <= 1e-11
    if cls in ("Lagrange", "DiscontinuousLagrange"):
        assert rel(el.V, g[key + "_V"]) <= 1e-12


def test_augmented_space_dimensions(oracle_backend):  # noqa: F811
    """dim RT_q = q (q + 2) / q (q+1)(q+3)/2, dim Ned_q = q (q + 2) / q (q+2)(q+3)/2 -- the SVD rank must hit them."""
    import fiat_amd
    from fiat_amd.nedelec import NedelecSpace2D, NedelecSpace3D
    from fiat_amd.raviart_thomas import RTSpace
    tri, tet = fiat_amd.ufc_simplex(2), fiat_amd.ufc_simplex(3)
    for q in (1, 2, 3, 4):
        assert len(RTSpace(tri, q)) == q * (q + 2) and len(NedelecSpace2D(tri, q)) == q * (q + 2)
        assert len(RTSpace(tet, q)) == q * (q + 1) * (q + 3) // 2
        assert len(NedelecSpace3D(tet, q)) == q * (q + 2) * (q + 3) // 2
    with pytest.raises(ValueError):
        NedelecSpace2D(tet, 1)
    with pytest.raises(ValueError):
        NedelecSpace3D(tri, 1)
