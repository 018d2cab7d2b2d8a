"""Wider families on the same coeffs x Dubiner kernels: Brezzi-Douglas-Marini
(FIAT/brezzi_douglas_marini.py), second-kind Nedelec (FIAT/nedelec_second_kind.py), Regge / Hellan-Herrmann-Johnson and
-- for the derivative functionals of FIAT/dual_set.py:175-205 -- cubic Hermite (FIAT/hermite.py) and Morley
(FIAT/morley.py), against golden vectors produced by the reference itself (tests/golden/make_golden_families.py)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("bdm", "BrezziDouglasMarini", 2, 1), ("bdm", "BrezziDouglasMarini", 2, 2), ("bdm", "BrezziDouglasMarini", 2, 3),
         ("bdm", "BrezziDouglasMarini", 3, 1), ("bdm", "BrezziDouglasMarini", 3, 2),
         ("n2curl", "NedelecSecondKind", 2, 1), ("n2curl", "NedelecSecondKind", 2, 2), ("n2curl", "NedelecSecondKind", 2, 3),
         ("n2curl", "NedelecSecondKind", 3, 1), ("n2curl", "NedelecSecondKind", 3, 2),
         # dual sets with derivative functionals (FIAT/dual_set.py:175-205)
         ("hermite", "CubicHermite", 1, 3), ("hermite", "CubicHermite", 2, 3), ("hermite", "CubicHermite", 3, 3),
         ("morley", "Morley", 2, 2), ("morley", "Morley", 3, 2)]
VECTOR = ("bdm", "n2curl")


def _pt(cls):
    return lambda fa, c, k: getattr(fa, cls)(c, k, variant="point")


# (golden name, constructor(fiat_amd, cell, degree), sd, degree): point variants and matrix-valued elements
MORE = [("rtpt", _pt("RaviartThomas"), 2, 1), ("rtpt", _pt("RaviartThomas"), 2, 2), ("rtpt", _pt("RaviartThomas"), 3, 1),
        ("rtpt", _pt("RaviartThomas"), 3, 2), ("nedpt", _pt("Nedelec"), 2, 1), ("nedpt", _pt("Nedelec"), 2, 2),
        ("nedpt", _pt("Nedelec"), 3, 1), ("nedpt", _pt("Nedelec"), 3, 2), ("bdmpt", _pt("BrezziDouglasMarini"), 2, 1),
        ("bdmpt", _pt("BrezziDouglasMarini"), 2, 2), ("bdmpt", _pt("BrezziDouglasMarini"), 3, 1),
        ("bdmpt", _pt("BrezziDouglasMarini"), 3, 2), ("n2curlpt", _pt("NedelecSecondKind"), 2, 1),
        ("n2curlpt", _pt("NedelecSecondKind"), 2, 2), ("n2curlpt", _pt("NedelecSecondKind"), 3, 1),
        ("n2curlpt", _pt("NedelecSecondKind"), 3, 2),
        ("regge", lambda fa, c, k: fa.Regge(c, k), 2, 0), ("regge", lambda fa, c, k: fa.Regge(c, k), 2, 1),
        ("regge", lambda fa, c, k: fa.Regge(c, k), 2, 2), ("regge", lambda fa, c, k: fa.Regge(c, k), 3, 0),
        ("regge", lambda fa, c, k: fa.Regge(c, k), 3, 1), ("reggept", lambda fa, c, k: fa.Regge(c, k, variant="point"), 2, 1),
        ("reggept", lambda fa, c, k: fa.Regge(c, k, variant="point"), 3, 1),
        ("hhj", lambda fa, c, k: fa.HellanHerrmannJohnson(c, k), 2, 0), ("hhj", lambda fa, c, k: fa.HellanHerrmannJohnson(c, k), 2, 1),
        ("hhj", lambda fa, c, k: fa.HellanHerrmannJohnson(c, k), 2, 2), ("hhj", lambda fa, c, k: fa.HellanHerrmannJohnson(c, k), 3, 0),
        ("hhj", lambda fa, c, k: fa.HellanHerrmannJohnson(c, k), 3, 1),
        ("hhjpt", lambda fa, c, k: fa.HellanHerrmannJohnson(c, k, variant="point"), 3, 1)]


def rel(x, ref):
    return np.abs(x - ref).max() / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,cls,sd,k", CASES)
def test_family_against_the_reference(golden, name, cls, sd, k):
    import fiat_amd
    g = golden("families")
    key = f"{name}{k}_sd{sd}"
    el = getattr(fiat_amd, cls)(fiat_amd.ufc_simplex(sd), k)
    co = g[key + "_coeffs"]
    assert el.get_coeffs().shape == co.shape
    assert rel(el.get_coeffs(), co) <= 1e-12, rel(el.get_coeffs(), co)
    assert el.mapping()[0] == str(g[key + "_mapping"]) and len(el.mapping()) == el.space_dimension()
    want = json.loads(str(g[key + "_entity_dofs"]))
    got = {str(d): {str(i): list(v) for i, v in ents.items()} for d, ents in el.entity_dofs().items()}
    assert got == want
    pts = g[f"pts_sd{sd}"]
    tab = el.tabulate(1, pts)
    alphas = [a for j in range(2) for a in fiat_amd.mis(sd, j)]
    assert list(tab) == alphas
    for t, a in enumerate(alphas):
        assert tab[a].shape == g[key + "_tab"][t].shape
        assert rel(tab[a], g[key + "_tab"][t]) <= (1e-12 if t == 0 else 1e-10), (a, rel(tab[a], g[key + "_tab"][t]))
    # batched path, the same points as two requests
    dev = el.tabulate_batch(1, np.stack([pts, pts[::-1]])).cpu().numpy()
    assert rel(dev[0], g[key + "_tab"]) <= 1e-10
    assert rel(dev[1][..., ::-1], g[key + "_tab"]) <= 1e-10
    if name in VECTOR:
        assert el.value_shape() == (sd,) and el.get_formdegree() == (sd - 1 if name == "bdm" else 1)
    else:
        assert el.value_shape() == ()


def test_family_errors():
    import fiat_amd
    with pytest.raises(Exception):
        fiat_amd.BrezziDouglasMarini(fiat_amd.ufc_simplex(2), 0)
    assert fiat_amd.supported_elements["Brezzi-Douglas-Marini"] is fiat_amd.BrezziDouglasMarini
    assert fiat_amd.supported_elements["Nedelec 2nd kind H(curl)"] is fiat_amd.NedelecSecondKind
    with pytest.raises(ValueError):
        fiat_amd.Morley(fiat_amd.ufc_simplex(2), 3)
    with pytest.raises(ValueError):
        fiat_amd.Morley(fiat_amd.ufc_simplex(1))


@pytest.mark.parametrize("npts", [9, 23])
def test_pushforward_of_the_new_families(golden, npts):
    """BDM / N2curl with their Piola maps on physical cells: fused or second-pass result == formula on
    the reference tables (the maps themselves are pinned in test_gpu_pushforward.py)."""
    import fiat_amd
    rng = np.random.default_rng(3)
    sd, nreq = 3, 40   # (23 points: stacked-matrix kernel + table-mixing pass + Piola pass)
    ref = np.array(fiat_amd.ufc_simplex(sd).get_vertices(), dtype=float)
    A = np.eye(sd) + 0.15 * rng.standard_normal((nreq, sd, sd))
    verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((nreq, 1, sd))
    e = rng.exponential(size=(nreq, npts, sd + 1))
    bary = e / e.sum(axis=-1, keepdims=True)
    pts = np.einsum("rpv,rvd->rpd", bary, verts)
    ref_pts = np.einsum("rpv,vd->rpd", bary, ref)
    for cls, deg in ((fiat_amd.BrezziDouglasMarini, 2), (fiat_amd.NedelecSecondKind, 2), (fiat_amd.Nedelec, 3)):
        el = cls(fiat_amd.ufc_simplex(sd), deg)
        got = el.tabulate_batch(1, pts, verts=verts, pushforward=True).cpu().numpy()
        raw = el.tabulate_batch(1, ref_pts).cpu().numpy()  # reference cell: (nreq, 4, ndof, sd, npts)
        for i in (0, 17, 39):
            J = (verts[i][1:] - verts[i][0]).T @ np.linalg.inv((ref[1:] - ref[0]).T)
            Jinv = np.linalg.inv(J)
            M = Jinv.T if el.mapping()[0].startswith("cov") else J / np.linalg.det(J)
            vals = np.einsum("ce,dep->dcp", M, raw[i, 0])
            grads = np.einsum("ce,gdep,gh->hdcp", M, raw[i, 1:], Jinv)
            assert rel(got[i, 0], vals) <= 1e-12
            assert rel(got[i, 1:], grads) <= 1e-10


@pytest.mark.parametrize("cls,degree", [("Lagrange", 7), ("DiscontinuousLagrange", 7)])
def test_large_vandermonde_systems_nodality(cls, degree):
    """120 dofs on a tetrahedron: V and the right-hand sides exceed the LDS, the solver works in a global
    workspace (vandermonde_solve_kernel<true>).  Nodal basis: phi_j(x_i) = delta_ij
    (test/FIAT/unit/test_fiat.py:76-117, test_nodality)."""
    import fiat_amd
    el = getattr(fiat_amd, cls)(fiat_amd.ufc_simplex(3), degree)
    assert el.space_dimension() == 120
    nodes = np.array([list(ell.get_point_dict().keys())[0] for ell in el.dual_basis()])
    tab = el.tabulate(0, nodes)[(0, 0, 0)]
    assert np.abs(tab - np.eye(120)).max() < 1e-9


@pytest.mark.parametrize("name,make,sd,k", MORE, ids=[f"{m[0]}{m[3]}_sd{m[2]}" for m in MORE])
def test_point_variants_and_matrix_valued_elements(golden, name, make, sd, k):
    """Point variants of RT / Nedelec / BDM / N2curl (normal and tangential point evaluations,
    FIAT/functional.py:499-614) and the symmetric-matrix-valued Regge / Hellan-Herrmann-Johnson elements against the
    reference."""
    import fiat_amd
    g = golden("families")
    key = f"{name}{k}_sd{sd}"
    el = make(fiat_amd, fiat_amd.ufc_simplex(sd), k)
    co = g[key + "_coeffs"]
    assert el.get_coeffs().shape == co.shape
    assert rel(el.get_coeffs(), co) <= 1e-11, rel(el.get_coeffs(), co)
    assert el.mapping()[0] == str(g[key + "_mapping"])
    want = json.loads(str(g[key + "_entity_dofs"]))
    got = {str(d): {str(i): list(v) for i, v in ents.items()} for d, ents in el.entity_dofs().items()}
    assert got == want
    tab = el.tabulate(1, g[f"pts_sd{sd}"])
    for t, a in enumerate([a for j in range(2) for a in fiat_amd.mis(sd, j)]):
        assert rel(tab[a], g[key + "_tab"][t]) <= (1e-11 if t == 0 else 1e-10), (a, rel(tab[a], g[key + "_tab"][t]))


def test_dual_set_restriction_indices():
    import fiat_amd
    el = fiat_amd.Lagrange(fiat_amd.ufc_simplex(2), 2)
    assert el.get_dual_set().get_indices("vertex") == [0, 1, 2]
    assert el.get_dual_set().get_indices("interior") == []
    assert el.get_dual_set().get_indices("edge", take_closure=False) == [3, 4, 5]
    with pytest.raises(RuntimeError):
        el.get_dual_set().get_indices("nonsense")


@pytest.mark.parametrize("name,cls", [("regge", "Regge"), ("hhj", "HellanHerrmannJohnson")])
@pytest.mark.parametrize("sd", [2, 3])
def test_double_piola_pushforward(golden, name, cls, sd):
    """Matrix-valued elements pushed forward to physical cells (double covariant J^-T Phi J^-1 for Regge, double
    contravariant J Phi J^T / det^2 for HHJ; fx_pushforward_batch kinds 3 / 4) equal the reference's elements
    built directly on those cells (one of them negatively oriented)."""
    import fiat_amd
    g = golden("families")
    verts, pts = g[f"phys_verts_sd{sd}"], g[f"phys_pts_sd{sd}"]
    el = getattr(fiat_amd, cls)(fiat_amd.ufc_simplex(sd), 1)
    assert el.mapping()[0] == ("double covariant piola" if name == "regge" else "double contravariant piola")
    out = el.tabulate_batch(1, pts, verts=verts, pushforward=True).cpu().numpy()
    want = g[f"{name}1_phys_sd{sd}_tab"]
    assert out.shape == want.shape
    for t in range(want.shape[1]):
        assert rel(out[:, t], want[:, t]) <= (1e-11 if t == 0 else 1e-10), (t, rel(out[:, t], want[:, t]))
    # errors: a vector-valued element cannot take a double map and vice versa
    rt = fiat_amd.RaviartThomas(fiat_amd.ufc_simplex(sd), 1).device_polyset()
    with pytest.raises(ValueError):
        rt.tabulate_batch(0, pts, verts=verts, mapping="double covariant piola")
    with pytest.raises(ValueError):
        el.device_polyset().tabulate_batch(0, pts, verts=verts, mapping="covariant piola")


def test_rotated_regge_is_hhj_on_the_device():
    """test/FIAT/unit/test_regge_hhj.py:7-19: on the triangle, the lowest-order Regge basis function r_i is S(h_i) of the
    Hellan-Herrmann-Johnson function h_i, S(u) = tr(u) I - u; single-point tabulation on the device, numpy.isclose as there."""
    import fiat_amd
    triangle = fiat_amd.UFCTriangle()
    R = fiat_amd.Regge(triangle, 0)
    H = fiat_amd.HellanHerrmannJohnson(triangle, 0)
    rt, ht = R.tabulate(0, (0.2, 0.2))[(0, 0)], H.tabulate(0, (0.2, 0.2))[(0, 0)]
    assert rt.shape == ht.shape == (3, 2, 2)
    for r, h in zip(rt, ht):
        assert np.all(np.isclose(r, np.eye(2) * np.trace(h) - h))
