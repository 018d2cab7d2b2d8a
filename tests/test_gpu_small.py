"""The lane-local kernel of the low-order simplex elements (simplex_small.hpp) against the C oracle:
every registered (sd, degree), orders 0 and 1, ragged batches, point counts from 1 to 64, with and
without per-request cells, scalar and vector-valued coefficient matrices."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu
SHAPES = [(2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (2, 4)]


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from fiat_amd import runtime
    runtime.Context.get()
    return runtime


def simplex_points(rng, sd, shape):
    e = rng.exponential(size=tuple(shape) + (sd + 1,))
    return (e / e.sum(axis=-1, keepdims=True))[..., 1:].copy()


def compare(out, ref, order):
    ref = ref.reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    num = np.abs(out - ref).max(axis=axes)
    den = np.maximum(1.0, np.abs(ref).max(axis=axes))
    err = (num / den).max(axis=0)
    assert err[0] <= 1e-12, err
    if order:
        assert err[1:].max() <= 1e-10, err   # (first and second derivatives)


@pytest.mark.parametrize("sd,n", SHAPES)
@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("nreq,npts", [(1, 1), (7, 3), (1000, 4), (333, 11), (129, 17), (65, 64), (4097, 6)])
@pytest.mark.parametrize("cells", [False, True])
def test_small_kernel_vs_c_oracle(rt, sd, n, order, nreq, npts, cells):
    from oracle import c_oracle
    import math
    rng = np.random.default_rng(1000 * sd + 100 * n + 10 * order + nreq + npts)
    nexp = math.comb(n + sd, sd)
    ndof = nexp + 3                                       # a non-square coefficient matrix
    co = rng.standard_normal((ndof, nexp))
    ps = rt.SimplexPolySet(sd, n, coeffs=co)
    ref_cell = fo.UFC_SIMPLEX[sd]
    verts = None
    pts = simplex_points(rng, sd, (nreq, npts))
    if cells:
        A = np.eye(sd) + 0.15 * rng.standard_normal((nreq, sd, sd))
        verts = np.einsum("vd,red->rve", ref_cell, A) + rng.standard_normal((nreq, 1, sd))
        e = rng.exponential(size=(nreq, npts, sd + 1))
        pts = np.einsum("rpv,rvd->rpd", e / e.sum(axis=-1, keepdims=True), verts)
    if 12 * 1024 >= 8 * (1 + sd * order) * ndof * npts or npts * 1 <= 64:
        assert ps.kernel_name(order, nreq, npts, has_verts=cells) in ("fxk::tabulate_simplex_small", "fxk::tabulate_simplex_kernel",
                                                                      "fxk::tabulate_simplex_stacked")
    out = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    ref = c_oracle.tabulate_batch(ref_cell, n, co, order, pts, verts=verts)
    compare(out, ref, order)


@pytest.mark.parametrize("sd,n", SHAPES)
def test_small_kernel_is_selected(rt, sd, n):
    import math
    nexp = math.comb(n + sd, sd)
    ps = rt.SimplexPolySet(sd, n, coeffs=np.eye(nexp))
    # shapes the stacked-matrix kernel also serves (round 2: degree-2 tetrahedra, degree-3 / 4 triangles) go there once the
    # derivative tables give it 16+ stacked rows and the points fill its column tiles (tools/small_vs_stacked.py)
    stacked_too = (sd, n) in ((3, 2), (2, 3), (2, 4))
    small, stacked, generic = ("fxk::tabulate_simplex_" + k for k in ("small", "stacked", "kernel"))
    assert ps.kernel_name(0, 1000, 4) == small
    assert ps.kernel_name(1, 1000, 4) == (stacked if (sd, n) == (3, 2) else small)    # twelve 4-point requests per group
    assert ps.kernel_name(0, 1000, 11, has_verts=True) == small
    assert ps.kernel_name(1, 1000, 11, has_verts=True) in ((small, stacked) if stacked_too else (small,))
    assert ps.kernel_name(2, 1000, 2) == small                               # Hessians too
    assert ps.kernel_name(1, 1000, 65) == (stacked if stacked_too else generic)  # more points than lanes


def test_small_kernel_vector_valued_and_bubble(rt, golden):
    """Through the facade: N1 / RT1 (vector valued) and P2 Lagrange (bubble variant, C0 transform folded in)."""
    import fiat_amd
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    for fam, sd, deg in (("Nedelec", 3, 1), ("RaviartThomas", 2, 1), ("Lagrange", 3, 2), ("Lagrange", 2, 3)):
        el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
        pts = simplex_points(rng, sd, (501, 7))
        # (P3 triangles at 7 points on their own cell: lane-local since round 3, plan_launch)
        assert el.device_polyset().kernel_name(1, 501, 7) == ("fxk::tabulate_simplex_small" if deg == 1 or (sd, deg) == (2, 3)
                                                              else "fxk::tabulate_simplex_stacked")
        out = el.tabulate_batch(1, pts).cpu().numpy()
        ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[sd], el.get_nodal_basis().get_embedded_degree() if hasattr(el.get_nodal_basis(), "get_embedded_degree") else deg,
                                      el.get_coeffs(), 1, pts, scale=el._expansion_scale, variant=el._expansion_variant)
        compare(out, ref, 1)
