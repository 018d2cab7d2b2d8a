"""CPU checks of the FInAT-side adapter's HOST logic against tests/golden/finat.npz (generated from the unmodified
reference by tests/golden/make_golden_finat.py): the dual-basis weight tensors Q and their unique points
(finat/fiat_elements.py:163-262), the run-time-tabulated argument names and shapes (finat/runtime_tabulated.py:68-95),
the multi-index split of tensor products (finat/tensor_product.py:98-144) and entity_support_dofs
(FIAT/finite_element.py:222-264, finat/finiteelementbase.py:85-119) -- with the oracle standing in for the device
arithmetic (tests/host_backend.py).  The same cases run on the HIP path in test_gpu_finat_adapter.py."""
import numpy as np
import pytest

from host_backend import oracle_backend  # noqa: F401  (fixture)

DUAL = [("P2tri", "Lagrange", 2, 2), ("P3tet", "Lagrange", 3, 3), ("DG2tet", "DiscontinuousLagrange", 3, 2),
        ("RT2tri", "RaviartThomas", 2, 2), ("N2tet", "Nedelec", 3, 2), ("BDM1tet", "BrezziDouglasMarini", 3, 1),
        ("Regge1tri", "Regge", 2, 1), ("RT2tet", "RaviartThomas", 3, 2)]
ESD = [("P3tet", "Lagrange", 3, 3), ("N2tet", "Nedelec", 3, 2), ("RT2tet", "RaviartThomas", 3, 2),
       ("DG2tet", "DiscontinuousLagrange", 3, 2), ("P2tri", "Lagrange", 2, 2), ("RT1tri", "RaviartThomas", 2, 1),
       ("BDM1tet", "BrezziDouglasMarini", 3, 1), ("P1int", "Lagrange", 1, 1)]


def esd_expected(g, prefix):
    return {str(k): g[f"{prefix}_{k}"].tolist() for k in g[f"{prefix}_keys"]}


def esd_flat(result_by_dim):
    out = {}
    for dim, ents in result_by_dim.items():
        tag = "-".join(map(str, dim)) if isinstance(dim, tuple) else str(dim)
        for f, dofs in ents.items():
            out[f"{tag}_{f}"] = list(dofs)
    return out


@pytest.mark.parametrize("name,cls,sd,degree", DUAL, ids=[d[0] for d in DUAL])
def test_dual_basis_equals_reference(golden, oracle_backend, name, cls, sd, degree):  # noqa: F811
    import fiat_amd
    from fiat_amd import finat_adapter as ad
    g = golden("finat")
    fe = ad.FiatElement(getattr(fiat_amd, cls)(fiat_amd.ufc_simplex(sd), degree))
    Q, ps = fe.dual_basis
    np.testing.assert_allclose(ps.points, g[f"db_{name}_pts"], rtol=0, atol=1e-14)
    assert Q.shape == g[f"db_{name}_Q"].shape
    np.testing.assert_allclose(Q, g[f"db_{name}_Q"], rtol=0, atol=1e-13)
    assert fe.Q_is_identity == bool(g[f"db_{name}_identity"])


def test_dual_basis_rejects_derivative_nodes_like_the_reference(golden, oracle_backend):  # noqa: F811
    import fiat_amd
    from fiat_amd import finat_adapter as ad
    assert int(golden("finat")["db_hermite_raises"]) == 1
    with pytest.raises(NotImplementedError):
        ad.FiatElement(fiat_amd.CubicHermite(fiat_amd.ufc_simplex(2))).dual_basis


def test_runtime_tabulated_names_and_shapes(golden):
    import fiat_amd
    from fiat_amd import finat_adapter as ad
    g = golden("finat")
    cell = fiat_amd.ufc_simplex(1)
    names, shapes = [], []
    for degree, kw, order, npts in ((3, dict(variant="equispaced", shift_axes=1, restriction='+', continuous=True), 2, 5),
                                    (2, dict(variant="gll", shift_axes=0, restriction=None, continuous=False), 1, 4),
                                    (1, dict(variant="spectral", shift_axes=2, restriction='-', continuous=True), 0, 3)):
        rt = ad.RuntimeTabulated(cell, degree, **kw)
        ps = ad.PointSet(np.linspace(0.1, 0.9, npts)[:, None])
        for arg in rt.basis_evaluation(order, ps).values():
            names.append(arg.name)
            shapes.append(list(arg.shape) + [rt.formdegree, rt.space_dimension()])
    assert names == [str(n) for n in g["rt_names"]]
    assert shapes == g["rt_shapes"].tolist()


def test_tensor_product_multiindex_split(golden, oracle_backend):  # noqa: F811
    import fiat_amd
    from fiat_amd import finat_adapter as ad
    g = golden("finat")
    cell = fiat_amd.ufc_simplex(1)
    tp = ad.TensorProductElement([ad.FiatElement(fiat_amd.Lagrange(cell, 2)), ad.FiatElement(fiat_amd.Lagrange(cell, 3)),
                                  ad.FiatElement(fiat_amd.DiscontinuousLagrange(cell, 1))])
    assert list(tp.index_shape) == g["tp_index_shape"].tolist()
    deltas = tp.factor_multiindices(1)
    assert [list(d) for d in deltas] == g["tp_deltas"].tolist()
    for Delta, parts in deltas.items():
        assert tuple(x for p in parts for x in p) == Delta


@pytest.mark.parametrize("name,cls,sd,degree", ESD, ids=[e[0] for e in ESD])
def test_entity_support_dofs_host(golden, oracle_backend, name, cls, sd, degree):  # noqa: F811
    import fiat_amd
    from fiat_amd.finite_element import entity_support_dofs
    g = golden("finat")
    el = getattr(fiat_amd, cls)(fiat_amd.ufc_simplex(sd), degree)
    got = esd_flat({dim: entity_support_dofs(el, dim) for dim in range(sd + 1)})
    assert got == esd_expected(g, f"esd_fiat_{name}")
    assert got == esd_expected(g, f"esd_finat_{name}")      # the FInAT twin computes the same supports
    assert entity_support_dofs(el, sd) is entity_support_dofs(el, sd)   # cached per element, like the reference


def test_tensor_product_degree_is_the_maximum(golden, oracle_backend):  # noqa: F811
    """FIAT/tensor_product.py:215-219: polydegree = max over the factors (entity_support_dofs derives its rule from it)."""
    import fiat_amd
    prism = fiat_amd.TensorProductElement(fiat_amd.Lagrange(fiat_amd.ufc_simplex(2), 2), fiat_amd.Lagrange(fiat_amd.ufc_simplex(1), 1))
    assert prism.degree() == int(golden("finat")["esd_prism_degree"]) == 2
