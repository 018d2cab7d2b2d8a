"""Push-forward of tabulated tables to physical cells through the C ABI.

Golden vectors (tests/golden/piola.npz, made by make_golden_piola.py from the unmodified
reference): Nedelec / Raviart-Thomas / Lagrange elements constructed DIRECTLY on random
physical simplices (one of them negatively oriented).  Where every degree of freedom
transforms with the Piola map (degree 1, and Nedelec degree 2 on tetrahedra: edge and face
tangent moments only) the pushed-forward reference basis must equal the reference's
physical-cell basis; elements with interior moments against Cartesian test vectors (N2 on
triangles, RT2) span the same space with a different interior basis, they are checked against
the oracle's evaluation of the formula  phi = M Phi(X(x)),  M = J^-T  or  J / det J."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu
TOL_VAL, TOL_DER = 1e-12, 1e-10


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from fiat_amd import runtime
    runtime.Context.get()
    return runtime


def check(got, ref, what):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    for t in range(ref.shape[0]):
        err = np.abs(got[t] - ref[t]).max() / max(1.0, np.abs(ref[t]).max())
        assert err <= (TOL_VAL if t == 0 else TOL_DER), (what, t, err)


CASES = [("n1", 1, "covariant piola", True), ("n2", 2, "covariant piola", None), ("rt1", 1, "contravariant piola", True),
         ("rt2", 2, "contravariant piola", False)]


@pytest.mark.parametrize("sd", [2, 3])
@pytest.mark.parametrize("name,n,mapping,exact", CASES)
def test_piola_pushforward(rt, golden, sd, name, n, mapping, exact):
    g = golden("piola")
    verts, pts = g[f"verts_sd{sd}"], g[f"pts_sd{sd}"]
    co = g[f"{name}_sd{sd}_refcoeffs"]
    if exact is None:
        exact = sd == 3          # N2: tetrahedra have no interior moments at degree 2, triangles do
    ps = rt.SimplexPolySet(sd, n, coeffs=co, value_shape=(sd,))
    out = ps.tabulate_batch(1, pts, verts=verts, mapping=mapping).cpu().numpy()
    ref_cell = fo.UFC_SIMPLEX[sd]
    for i in range(verts.shape[0]):
        # the formula, evaluated by the oracle
        J = (verts[i][1:] - verts[i][0]).T @ np.linalg.inv((ref_cell[1:] - ref_cell[0]).T)
        M = np.linalg.inv(J).T if mapping.startswith("cov") else J / np.linalg.det(J)
        tab = fo.element_tabulate(verts[i], n, co, 1, pts[i])
        raw = np.stack([tab[a] for a in fo.jet_indices(sd, 1)])
        check(out[i], np.einsum("ce,tdep->tdcp", M, raw), f"{name} sd{sd} cell {i} vs formula")
        if exact:
            check(out[i], g[f"{name}_sd{sd}_tab"][i], f"{name} sd{sd} cell {i} vs the reference on the physical cell")


@pytest.mark.parametrize("sd", [2, 3])
def test_affine_pullback_is_the_physical_element(rt, golden, sd):
    """Lagrange P2: reference coefficients + cell geometry == the reference's element on the physical cell."""
    g = golden("piola")
    ps = rt.SimplexPolySet(sd, 2, variant="bubble", scale=1, coeffs=g[f"p2_sd{sd}_refcoeffs"])
    out = ps.tabulate_batch(1, g[f"pts_sd{sd}"], verts=g[f"verts_sd{sd}"], mapping="affine").cpu().numpy()
    for i in range(out.shape[0]):
        check(out[i], g[f"p2_sd{sd}_tab"][i], f"P2 sd{sd} cell {i}")


def test_pushforward_errors(rt, golden):
    g = golden("piola")
    ps = rt.SimplexPolySet(3, 2, variant="bubble", scale=1, coeffs=g["p2_sd3_refcoeffs"])
    with pytest.raises(ValueError):          # scalar-valued element: no Piola map
        ps.tabulate_batch(1, g["pts_sd3"], verts=g["verts_sd3"], mapping="covariant piola")
    with pytest.raises(ValueError):          # needs the physical cells
        ps.tabulate_batch(1, g["pts_sd3"], mapping="contravariant piola")
    with pytest.raises(ValueError):
        ps.tabulate_batch(1, g["pts_sd3"], verts=g["verts_sd3"], mapping="double covariant piola")


def test_facade_pushforward(rt, golden):
    """Nedelec(ufc tet, 1).tabulate_batch(..., pushforward=True) equals the reference's Nedelec on the physical cell."""
    import fiat_amd
    g = golden("piola")
    el = fiat_amd.Nedelec(fiat_amd.ufc_simplex(3), 1)
    assert el.mapping()[0] == "covariant piola"
    out = el.tabulate_batch(1, g["pts_sd3"], verts=g["verts_sd3"], pushforward=True).cpu().numpy()
    for i in range(out.shape[0]):
        check(out[i], g["n1_sd3_tab"][i], f"facade N1 cell {i}")
    rtel = fiat_amd.RaviartThomas(fiat_amd.ufc_simplex(2), 1)
    out = rtel.tabulate_batch(1, g["pts_sd2"], verts=g["verts_sd2"], pushforward=True).cpu().numpy()
    for i in range(out.shape[0]):
        check(out[i], g["rt1_sd2_tab"][i], f"facade RT1 tri cell {i}")


# ---- one reference point set in many cells (fx_tabulate_batch_shared) -----------------
SHARED = [("p2", 2, "affine", "bubble", 1, ()), ("n1", 1, "covariant piola", None, None, "v"),
          ("n2", 2, "covariant piola", None, None, "v"), ("rt1", 1, "contravariant piola", None, None, "v"),
          ("rt2", 2, "contravariant piola", None, None, "v")]


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("sd", [2, 3])
@pytest.mark.parametrize("name,n,mapping,variant,scale,vs", SHARED)
def test_shared_points_equal_per_request_points(rt, golden, sd, order, name, n, mapping, variant, scale, vs):
    """fx_tabulate_batch_shared == fx_tabulate_batch(points = F_r(ref points), verts) + push-forward."""
    g = golden("piola")
    co = g[f"{name}_sd{sd}_refcoeffs"]
    kw = dict(coeffs=co)
    if variant:
        kw.update(variant=variant, scale=scale)
    if vs == "v":
        kw.update(value_shape=(sd,))
    ps = rt.SimplexPolySet(sd, n, **kw)
    rng = np.random.default_rng(31 + sd + order)
    nreq, npts = 37, 11
    ref = fo.UFC_SIMPLEX[sd]
    e = rng.exponential(size=(npts, sd + 1))
    bary = e / e.sum(axis=1, keepdims=True)
    ref_pts = bary @ ref
    A = np.eye(sd) + 0.2 * rng.standard_normal((nreq, sd, sd))
    A[3, :, 0] *= -1.0                                  # a negatively oriented cell
    verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((nreq, 1, sd))
    pts = np.einsum("pv,rvd->rpd", bary, verts)
    want = ps.tabulate_batch(order, pts, verts=verts, mapping=mapping).cpu().numpy()
    got = ps.tabulate_batch_shared(order, ref_pts, verts, mapping=mapping).cpu().numpy()
    assert got.shape == want.shape
    for t in range(want.shape[1]):
        err = np.abs(got[:, t] - want[:, t]).max() / max(1.0, np.abs(want[:, t]).max())
        assert err <= (1e-12 if t == 0 else 1e-10), (name, sd, order, t, err)


def test_shared_points_against_the_reference_on_physical_cells(rt, golden):
    """N2 on tetrahedra through the facade: the shared-point path reproduces the reference's Nedelec
    constructed directly on each physical cell, at that cell's image of the shared points."""
    import fiat_amd
    g = golden("piola")
    el = fiat_amd.Nedelec(fiat_amd.ufc_simplex(3), 2)
    verts = g["verts_sd3"]
    ref = fo.UFC_SIMPLEX[3]
    # the golden points of cell 0, pulled back to the reference cell, serve as the shared set
    bary0 = np.linalg.solve(np.vstack([verts[0].T, np.ones(4)]), np.vstack([g["pts_sd3"][0].T, np.ones(7)])).T
    ref_pts = bary0 @ ref
    out = el.tabulate_cells(1, ref_pts, verts[:1]).cpu().numpy()
    check(out[0], g["n2_sd3_tab"][0], "shared points, N2 tet, cell 0")


@pytest.mark.parametrize("name,n,mapping", [("n2", 2, "covariant piola"), ("rt2", 2, "contravariant piola")])
@pytest.mark.parametrize("npts", [22, 23, 24])
@pytest.mark.parametrize("kernel", ["wave", "cooperative"])
def test_fused_pushforward_equals_second_pass(rt, golden, kernel_policy, name, n, mapping, npts, kernel):
    """N2 / RT2 tetrahedra at benchmark-like sizes: the push-forward fused into the kernel (applied to
    the LDS image of the one-request-per-wave kernel; in the output rounds of the cooperative kernel,
    several rounds per request, odd table sizes for RT2 at 23 points) against tabulation followed by
    the separate pass, and both against the oracle's evaluation of the formula."""
    if kernel == "cooperative":
        kernel_policy("no_fixed")
    g = golden("piola")
    co = g[f"{name}_sd3_refcoeffs"]
    ps = rt.SimplexPolySet(3, n, coeffs=co, value_shape=(3,))
    rng = np.random.default_rng(77 + npts)
    nreq = 301
    ref = fo.UFC_SIMPLEX[3]
    A = np.eye(3) + 0.15 * rng.standard_normal((nreq, 3, 3))
    A[5, :, 0] *= -1.0
    verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((nreq, 1, 3))
    e = rng.exponential(size=(nreq, npts, 4))
    pts = np.einsum("rpv,rvd->rpd", e / e.sum(axis=-1, keepdims=True), verts)
    fused = ps.tabulate_batch(1, pts, verts=verts, mapping=mapping).cpu().numpy()
    two = ps.pushforward_batch(1, ps.tabulate_batch(1, pts, verts=verts), verts, mapping).cpu().numpy()
    for t in range(4):
        err = np.abs(fused[:, t] - two[:, t]).max() / max(1.0, np.abs(two[:, t]).max())
        assert err <= 1e-13, (name, npts, t, err)
    for i in (0, 5, 150, 300):
        J = (verts[i][1:] - verts[i][0]).T @ np.linalg.inv((ref[1:] - ref[0]).T)
        M = np.linalg.inv(J).T if mapping.startswith("cov") else J / np.linalg.det(J)
        tab = fo.element_tabulate(verts[i], n, co, 1, pts[i])
        raw = np.stack([tab[a] for a in fo.jet_indices(3, 1)])
        check(fused[i], np.einsum("ce,tdep->tdcp", M, raw), f"{name} fused, cell {i}")
