"""C1 macro spaces and the Hsieh-Clough-Tocher element on the device against golden
vectors produced by the reference itself (tests/golden/make_golden_macro.py::hct_main).

Reference: FIAT/macro.py:381-432 (MacroQuadratureRule), :435-521 (CkPolynomialSet), FIAT/expansions.py:492-574
(normal-derivative jumps), FIAT/hct.py:19-88.  Tolerances: 1e-12 on values, 1e-10 on derivatives and on
coefficients obtained through the SVD nullspace + Vandermonde solve."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu


def rel(x, ref):
    return np.max(np.abs(x - ref)) / max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("name,kw", [("hct3", dict(degree=3)), ("hct3_reduced", dict(degree=3, reduced=True)),
                                     ("hct4", dict(degree=4)), ("hct5", dict(degree=5))])
def test_hct_against_reference(golden, name, kw):
    import fiat_amd as fa
    G = golden("hct")
    e = fa.HsiehCloughTocher(fa.ufc_simplex(2), **kw)
    assert e.is_macroelement() and e.get_reference_complex().is_macrocell()
    ids = e.entity_dofs()
    flat = [(d, ent, dof) for d in sorted(ids) for ent in sorted(ids[d]) for dof in ids[d][ent]]
    assert np.array_equal(np.array(flat).reshape(-1, 3), G[f"{name}/entity_dofs"])
    # the nodal basis is unique although the SVD basis of the C1 space is not: compare the functions
    pts = G[f"{name}/pts"]
    tab = e.tabulate(2, pts)
    got = np.stack([tab[a] for a in fo.jet_indices(2, 2)])
    ref = G[f"{name}/tab2"]
    assert got.shape == ref.shape
    assert rel(got[0], ref[0]) <= 1e-11
    for t in range(1, 6):
        assert rel(got[t], ref[t]) <= 1e-10 * max(1.0, np.max(np.abs(ref[t]))), (t, rel(got[t], ref[t]))
    # C1: the gradient is single-valued across the interior edges -- evaluate on both sides of the barycentre spokes
    S = e.get_reference_complex()
    V = np.array(S.get_vertices())
    eps = 1e-7
    for k in range(3):
        mid = 0.5 * (V[k] + V[3])                       # point on the spoke from vertex k to the barycentre
        nrm = np.array([-(V[3] - V[k])[1], (V[3] - V[k])[0]])
        nrm /= np.linalg.norm(nrm)
        ta = e.tabulate(1, np.array([mid + eps * nrm]))
        tb = e.tabulate(1, np.array([mid - eps * nrm]))
        for a in [(1, 0), (0, 1)]:
            assert np.max(np.abs(ta[a] - tb[a])) < 1e-4 * max(1.0, np.max(np.abs(ta[a])))


@pytest.mark.parametrize("key,split,sd,deg", [("AlfeldSplit2", "AlfeldSplit", 2, 3), ("AlfeldSplit2", "AlfeldSplit", 2, 4),
                                              ("AlfeldSplit3", "AlfeldSplit", 3, 3), ("PowellSabinSplit2", "PowellSabinSplit", 2, 2)])
@pytest.mark.parametrize("variant", [None, "bubble"])
def test_ck_space_equals_reference(golden, key, split, sd, deg, variant):
    """The C1 space itself: same dimension and same span (orthogonal projector) as the reference's."""
    import fiat_amd as fa
    from fiat_amd import macro
    G = golden("hct")
    S = getattr(fa, split)(fa.ufc_simplex(sd))
    P = macro.CkPolynomialSet(S, deg, order=1, variant=variant)
    C = P.get_coeffs()
    ref = G[f"ck/{key}/deg{deg}/{variant or 'none'}/projector"]
    assert C.shape[0] == int(round(np.trace(ref)))
    proj = C.T @ np.linalg.solve(C @ C.T, C)
    assert np.max(np.abs(proj - ref)) < 1e-9


def test_macro_quadrature(golden):
    import fiat_amd as fa
    from fiat_amd import macro
    G = golden("hct")
    T = fa.ufc_simplex(2)
    Q = fa.create_quadrature(fa.AlfeldSplit(T), 3)
    x, w = Q.get_points(), Q.get_weights()
    assert abs(w.sum() - 0.5) < 1e-14
    # composite rule: exact for piecewise polynomials; compare integrals of monomials and, when the reference
    # uses the same rule on the sub-cells, the points themselves
    for a, b in [(0, 0), (1, 0), (1, 1), (2, 1), (0, 3)]:
        exact = np.dot(G["mq/alfeld_tri/wts"], G["mq/alfeld_tri/pts"][:, 0] ** a * G["mq/alfeld_tri/pts"][:, 1] ** b)
        assert abs(np.dot(w, x[:, 0] ** a * x[:, 1] ** b) - exact) < 1e-14
    Q = macro.MacroQuadratureRule(fa.IsoSplit(T), fa.create_quadrature(fa.ufc_simplex(1), 2), parent_facets=[0, 2])
    assert np.allclose(Q.get_points(), G["mq/iso_tri_facets/pts"], atol=1e-14)
    assert np.allclose(Q.get_weights(), G["mq/iso_tri_facets/wts"], atol=1e-14)


def test_hct_batch_physical_cells():
    """Batched tabulation of HCT on physical triangles: derivatives are physical ones == the same reference-cell
    coefficients evaluated by the oracle on the mapped split."""
    import fiat_amd as fa
    e = fa.HsiehCloughTocher(fa.ufc_simplex(2), 3)
    S = e.get_reference_complex()
    es = e.get_nodal_basis().get_expansion_set()
    rng = np.random.default_rng(3)
    nreq, npts = 50, 16
    ex = rng.exponential(size=(nreq, npts, 3))
    bary = ex / ex.sum(-1, keepdims=True)
    verts = np.array(S.get_parent().get_vertices())[None] + rng.uniform(-0.2, 0.2, size=(nreq, 3, 2))
    pts = np.einsum("rpk,rkd->rpd", bary, verts)
    out = e.tabulate_batch(2, pts, verts=verts).cpu().numpy()
    Vs = np.array(S.get_vertices())
    vb = np.concatenate([1.0 - Vs.sum(-1, keepdims=True), Vs], axis=-1)
    top = S.get_topology()
    cmap, coeffs = es.get_cell_node_map(3), e.get_coeffs()
    for r in range(nreq):
        pv = vb @ verts[r]
        cells = [pv[list(top[2][c])] for c in sorted(top[2])]
        ref = fo.macro_element_tabulate(verts[r], cells, cmap, 3, coeffs, 2, pts[r], es.scale, es.variant)
        for t, a in enumerate(fo.jet_indices(2, 2)):
            assert rel(out[r, t], ref[a]) <= (1e-12 if t == 0 else 1e-10)
