"""Round-4 parity cases on the HIP path.

* the request-per-workgroup kernel (fiat_amd/csrc/simplex_wg.hpp): rules of 49..128 points -- the 74- and 122-point rules
  of degree-5 / 6 tetrahedra (FIAT/xg_quad_data.py via FIAT/quadrature_schemes.py), the tabulation of
  FIAT/polynomial_set.py:68-72 with FIAT/expansions.py:140-267 underneath -- against the pinned C oracle at the north-star
  tolerances (1e-12 values, 1e-10 derivatives): every (degree, column-tile count) family, orders 0-2, odd table sizes (the
  8-byte flush twins), per-request cells (values: the cell only enters the production phase; derivatives behind policy
  no_stacked_mix: kernel + table-mixing pass), batches of 1 request, fewer requests than workgroups, and several requests
  per workgroup (the dynamic hand-out)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_VAL, TOL_DER = 1e-12, 1e-10


def rel(x, ref):
    return np.abs(x - ref).max() / max(1.0, np.abs(ref).max())


def oracle_tables(el, sd, order, pts, verts, shape):
    from oracle import c_oracle
    from oracle import fiat_oracle as fo
    n = el.get_nodal_basis().get_embedded_degree()
    return c_oracle.tabulate_batch(fo.UFC_SIMPLEX[sd], n, el.get_coeffs(), order, pts, verts=verts, scale=el._expansion_scale,
                                   variant=el._expansion_variant).reshape(shape)


def batch(sd, nreq, npts, seed, cells):
    from oracle import fiat_oracle as fo
    rng = np.random.default_rng(seed)
    e = rng.exponential(size=(nreq, npts, sd + 1))
    bary = e / e.sum(-1, keepdims=True)
    if not cells:
        return bary[..., 1:].copy(), None
    A = np.eye(sd) + 0.15 * rng.standard_normal((nreq, sd, sd))
    A[::3, :, 0] *= -1.0        # both orientations
    verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[sd], A) + rng.standard_normal((nreq, 1, sd))
    return np.einsum("rpv,rvd->rpd", bary, verts), verts


# (family, sd, degree, points) -> column tiles of the instance
WG = [("DiscontinuousLagrange", 3, 6, 122), ("Lagrange", 3, 6, 74), ("Lagrange", 3, 6, 57), ("Lagrange", 3, 6, 49), ("Lagrange", 3, 6, 128),
      ("Lagrange", 3, 6, 97), ("Lagrange", 3, 5, 74), ("Lagrange", 3, 5, 122), ("DiscontinuousLagrange", 3, 5, 111), ("Lagrange", 3, 5, 65),
      ("Lagrange", 3, 4, 70), ("Lagrange", 3, 4, 97), ("Lagrange", 3, 4, 122), ("Lagrange", 3, 3, 70), ("Lagrange", 3, 3, 97),
      ("Nedelec", 3, 3, 74), ("RaviartThomas", 3, 3, 81), ("Lagrange", 2, 6, 73), ("Lagrange", 2, 5, 79), ("Lagrange", 2, 6, 128),
      # odd table sizes: the 8-byte flush twins (35 x 97, 21 x 79 / 63 x 79 doubles a request ...)
      ("Lagrange", 3, 4, 97), ("Lagrange", 2, 5, 79), ("DiscontinuousLagrange", 2, 5, 65), ("RaviartThomas", 3, 3, 81),
      # one row tile per wave (values-only requests of up to 64 rows: the FAST instances, both flush widths; 35 rows: a wave without
      # a tile) and one wave per row tile with a last tile of 7 rows x 75 points (deferred like the full ones)
      ("Lagrange", 3, 4, 75), ("DiscontinuousLagrange", 3, 5, 80), ("Nedelec", 3, 3, 75)]


@pytest.mark.parametrize("family,sd,degree,npts", WG, ids=[f"{m[0][:3]}{m[2]}-sd{m[1]}-{m[3]}pt" for m in WG])
@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("nreq", [1, 77, 700])
def test_request_per_workgroup_kernel(family, sd, degree, npts, order, nreq, kernel_policy):
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    n = el.get_nodal_basis().get_embedded_degree()
    # 65..96 points with a short K loop (degree <= 4 tetrahedra, triangles except P6 with Hessians) stay on the point chunks by
    # default (the planner's measured rule); policy wg_small brings them here: every instance is tested either way
    # (... except where the five-tile instance -- one wave per row tile -- has no more MFMA slots per wave than the six-tile one)
    rt = -(-int(np.prod(ps.out_shape(order, 1, npts)[1:-1])) // 16)
    five = (npts + 15) // 16 == 5 and -(-rt // 4) * 5 <= -(-rt // 2) * 3
    if 64 < npts <= 96 and not ((sd == 3 and n >= 5) or (sd == 2 and n == 6 and order == 2) or five):
        assert "simplex_wg" not in ps.kernel_name(order, nreq, npts)
        kernel_policy("wg_small")
    name = ps.kernel_name(order, nreq, npts, instance=True)
    # column tiles of the instance: ceil(npts / 16), or one more where that layout has fewer MFMA slots per wave (5 -> 6 for
    # few row tiles) or no instance (7 -> 8)
    # (rules of 49..64 points: two requests share the slab)
    g = 2 if npts <= 64 else 1
    ct = (g * npts + 15) // 16
    sfx = "x2" if g == 2 else ""
    assert name in (f"fxk::tabulate_simplex_wg<{sd},{n},{ct}>{sfx}", f"fxk::tabulate_simplex_wg<{sd},{n},{ct + 1}>{sfx}") and "7>" not in name, name
    assert ps.kernel_name(order, nreq, npts) == "fxk::tabulate_simplex_wg"
    pts, _ = batch(sd, nreq, npts, 31 * npts + degree + order, False)
    got = ps.tabulate_batch(order, pts).cpu().numpy()
    ref = oracle_tables(el, sd, order, pts, None, got.shape)
    for t in range(got.shape[1]):
        assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (name, t, rel(got[:, t], ref[:, t]))
    # a second launch on the same context: the request counter of the first one cleaned up after itself
    again = ps.tabulate_batch(order, pts).cpu().numpy()
    assert np.array_equal(again, got)


@pytest.mark.parametrize("family,sd,degree,npts", [("Lagrange", 3, 6, 122), ("Lagrange", 3, 5, 74), ("Lagrange", 3, 4, 97), ("Lagrange", 2, 6, 73),
                                                   ("Nedelec", 3, 3, 74)])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_request_per_workgroup_kernel_with_cells(family, sd, degree, npts, order, kernel_policy):
    """Per-request cells: values straight from the kernel (the cell maps the points in the production phase); with
    derivatives the chain-rule instances of the stacked kernel own the shape by default, and policy no_stacked_mix sends it
    here + through the table-mixing pass."""
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    nreq = 333
    if order >= 1:   # default: the chain rule inside a kernel (order 1: this one, MIX; order 2: the stacked kernel's point chunks)
        name = ps.kernel_name(order, nreq, npts, has_verts=True, instance=True)
        assert ("+mix" in name) if order == 1 else ("stacked" in name), name
        kernel_policy("no_stacked_mix", "wg_small")     # (wg_small: past the planner's short-K rule for 65..96 points)
    else:
        kernel_policy("wg_small")
    assert ps.kernel_name(order, nreq, npts, has_verts=True) == "fxk::tabulate_simplex_wg"
    pts, verts = batch(sd, nreq, npts, 7 * npts + order, True)
    got = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    ref = oracle_tables(el, sd, order, pts, verts, got.shape)
    for t in range(got.shape[1]):
        assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (t, rel(got[:, t], ref[:, t]))


@pytest.mark.parametrize("npts,order", [(122, 2), (74, 1), (57, 0)])
def test_policy_no_wg_keeps_the_point_chunked_route(npts, order, kernel_policy):
    import fiat_amd as fa
    el = fa.Lagrange(fa.ufc_simplex(3), 6)
    ps = el.device_polyset()
    assert ps.kernel_name(order, 100, npts) == "fxk::tabulate_simplex_wg"
    pts, _ = batch(3, 100, npts, npts, False)
    got = ps.tabulate_batch(order, pts).cpu().numpy()
    kernel_policy("no_wg")
    assert ps.kernel_name(order, 100, npts) == "fxk::tabulate_simplex_stacked"
    want = ps.tabulate_batch(order, pts).cpu().numpy()
    ref = oracle_tables(el, 3, order, pts, None, got.shape)
    for t in range(got.shape[1]):
        tol = TOL_VAL if t == 0 else TOL_DER
        assert rel(got[:, t], ref[:, t]) <= tol and rel(want[:, t], ref[:, t]) <= tol


def test_rule_sizes_next_to_the_window():
    """48 points stay with the whole-request instances, 129 with the point chunks."""
    import fiat_amd as fa
    ps = fa.Lagrange(fa.ufc_simplex(3), 6).device_polyset()
    assert ps.kernel_name(1, 100, 48) == "fxk::tabulate_simplex_stacked"
    assert ps.kernel_name(1, 100, 129) == "fxk::tabulate_simplex_stacked"
    assert ps.kernel_name(1, 100, 128) == "fxk::tabulate_simplex_wg"


# (family, sd, degree, points) -> requests per slab, column tiles
WG_SMALL = [("Lagrange", 3, 6, 23, 5, 8), ("Lagrange", 3, 5, 14, 9, 8), ("DiscontinuousLagrange", 3, 5, 23, 5, 8), ("Lagrange", 2, 5, 25, 5, 8),
            ("Lagrange", 2, 6, 33, 3, 8), ("Lagrange", 3, 4, 44, 2, 6), ("Lagrange", 3, 6, 44, 2, 6), ("Lagrange", 3, 5, 31, 4, 8),
            ("Nedelec", 3, 3, 23, 5, 8), ("RaviartThomas", 3, 3, 14, 9, 8), ("Lagrange", 3, 3, 57, 2, 8), ("Lagrange", 3, 6, 64, 2, 8),
            ("Lagrange", 2, 5, 16, 8, 8), ("Lagrange", 3, 4, 11, 11, 8)]


@pytest.mark.parametrize("family,sd,degree,npts,g,ct", WG_SMALL, ids=[f"{m[0][:3]}{m[2]}-sd{m[1]}-{m[3]}pt" for m in WG_SMALL])
@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("nreq,cells", [(1, False), (7, False), (1031, False), (2400, False), (333, True)])
def test_groups_of_small_requests_per_workgroup(family, sd, degree, npts, g, ct, order, nreq, cells, kernel_policy):
    """Policy wg_small: several requests of <= 64 points share the slab of a workgroup (simplex_wg.hpp, gslab > 1): the image
    of a row tile is [request][16][npts], a request's rows leave as contiguous pieces.  Odd table sizes (8-byte twins), a last
    group with missing requests (1031 = 5 x 206 + 1 ...), fewer groups than workgroups, per-request cells (values from the
    kernel; derivatives + the table-mixing pass), against the C oracle."""
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    n = el.get_nodal_basis().get_embedded_degree()
    kernel_policy("wg_small", "no_fixed", "no_small", "no_stacked_mix")
    name = ps.kernel_name(order, nreq, npts, has_verts=cells, instance=True)
    if cells and order >= 1 and "simplex_wg" not in name:
        pytest.skip("per-request cells with derivatives: the planner keeps this shape off the two-pass routes (" + name + ")")
    assert name == f"fxk::tabulate_simplex_wg<{sd},{n},{ct}>x{g}", name
    pts, verts = batch(sd, nreq, npts, 17 * npts + degree + order + nreq, cells)
    got = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    ref = oracle_tables(el, sd, order, pts, verts, got.shape)
    for t in range(got.shape[1]):
        assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (name, t, rel(got[:, t], ref[:, t]))


# ---- derivative orders 5 and 6, on the element's cell and with per-request cells (tests/golden/round4.npz) ------------------
HO = [("dg6tet", 3, lambda fa, c: fa.DiscontinuousLagrange(c, 6), True), ("p6tri", 2, lambda fa, c: fa.Lagrange(c, 6), True),
      ("p5tet", 3, lambda fa, c: fa.Lagrange(c, 5), True), ("n4tri", 2, lambda fa, c: fa.Nedelec(c, 4), False),
      ("on7int", 1, lambda fa, c: fa.ONPolynomialSet(c, 7), False)]


def _chain_rule_tables(fa, ref_tab, sd, order, Kt):
    """d^alpha_x from the reference's tables with respect to X: Kt[c, d] = dX_c / dx_d, sum over ordered source directions
    (NumPy, independent of the device pass)."""
    import itertools
    keys = [a for k in range(order + 1) for a in fa.mis(sd, k)]
    pos = {a: i for i, a in enumerate(keys)}
    out = []
    for alpha in keys:
        dirs = [d for d, m in enumerate(alpha) for _ in range(m)]
        acc = 0.0
        for src in itertools.product(range(sd), repeat=len(dirs)):
            beta = tuple(src.count(c) for c in range(sd))
            acc = acc + float(np.prod([Kt[c, d] for c, d in zip(src, dirs)])) * ref_tab[pos[beta]]
        out.append(acc)
    return np.stack(out)


@pytest.mark.parametrize("name,sd,make,rebuild", HO, ids=[h[0] for h in HO])
@pytest.mark.parametrize("order", [5, 6])
def test_orders_5_and_6_vs_reference(golden, name, sd, make, rebuild, order):
    """Orders the earlier fixtures do not reach.  (a) On the element's own cell: `tabulate(order, points)` equals the
    reference's tables (FIAT/expansions.py:66-137 / :438-446).  (b) Per-request cells (`table_mix_any_kernel`, FIAT/expansions.py:
    411-447 through Jinv): equal to the reference's elements built ON the physical cells (affine families, incl. a negatively
    oriented cell), and to the chain rule applied in NumPy to the REFERENCE's reference-cell tables (all families).  Tolerance:
    the north-star 1e-10 for derivatives, relative to the largest entry of the table's order (sixth derivatives of degree-6
    bases reach 1e6)."""
    import fiat_amd as fa
    g = golden("round4")
    ref = np.array(fa.ufc_simplex(sd).get_vertices(), dtype=float)
    verts, pts, ref_pts = g[f"ho_{name}_verts"], g[f"ho_{name}_pts"], g[f"ho_{name}_refpts"]
    base = make(fa, fa.ufc_simplex(sd))
    is_element = hasattr(base, "dual_basis")
    dev = base if is_element else base.device_polyset()
    firsts = np.cumsum([0] + [len(fa.mis(sd, k)) for k in range(order + 1)])

    def close(got, want, tag):
        assert got.shape == want.shape, (tag, got.shape, want.shape)
        for k in range(order + 1):
            sl = slice(firsts[k], firsts[k + 1])
            err = np.abs(got[sl] - want[sl]).max() / max(1.0, np.abs(want[sl]).max())
            assert err <= (1e-12 if k == 0 else 1e-10), (tag, k, err)

    got = dev.tabulate_batch(order, ref_pts).cpu().numpy()
    cells = dev.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    for r in range(verts.shape[0]):
        want_ref = g[f"ho_{name}_o{order}_ref{r}"]
        close(got[r].reshape(want_ref.shape), want_ref, "own cell")
        J = (verts[r][1:] - verts[r][0]).T @ np.linalg.inv((ref[1:] - ref[0]).T)          # dx/dX
        close(cells[r].reshape(want_ref.shape), _chain_rule_tables(fa, want_ref, sd, order, np.linalg.inv(J)), "cells, chain rule")
        if rebuild:
            close(cells[r].reshape(want_ref.shape), g[f"ho_{name}_o{order}_phys{r}"], "cells, reference element on the physical cell")


def test_two_vector_valued_factors_are_refused_like_the_reference():
    """FIAT/tensor_product.py:268-271 raises NotImplementedError("tabulate does not support two vector-valued inputs"); so does
    the facade (the reference has no such tabulation to be equal to)."""
    import fiat_amd as fa
    tri, seg = fa.ufc_simplex(2), fa.ufc_simplex(1)
    # (the reference already fails while it builds the dual basis of such a product -- NotImplementedError("unsupported
    # functional type"), :99-205 -- so the facade refuses at construction, with the message of the tabulation)
    with pytest.raises(NotImplementedError, match="two vector-valued"):
        fa.TensorProductElement(fa.RaviartThomas(tri, 1), fa.RaviartThomas(tri, 1))
    ok = fa.TensorProductElement(fa.RaviartThomas(tri, 1), fa.Lagrange(seg, 1))     # one vector-valued factor: served
    assert ok.tabulate(0, np.array([[0.2, 0.3, 0.5]]))[(0, 0, 0)].shape == (6, 2, 1)


@pytest.mark.parametrize("family,sd,degree,npts", [("Lagrange", 2, 5, 7), ("Lagrange", 2, 5, 12), ("DiscontinuousLagrange", 2, 5, 25),
                                                   ("Lagrange", 2, 5, 30), ("Lagrange", 2, 5, 64), ("Lagrange", 3, 3, 4),
                                                   ("DiscontinuousLagrange", 3, 3, 11), ("Lagrange", 3, 3, 14)])
@pytest.mark.parametrize("nreq,cells", [(1, False), (515, False), (3, True), (1030, True)])
def test_values_only_requests_on_the_lane_local_kernel(family, sd, degree, npts, nreq, cells, kernel_policy):
    """Round 4: values-only requests of P5 triangles (21 rows) and of P3 tetrahedra up to the 14-point rule (20 rows) take the
    lane-local kernel (simplex_small.hpp: rows x members FMAs per lane with scalar coefficients) instead of MFMA tiles that are a
    third full; with derivatives the shapes stay where they were.  Against the C oracle; the former route behind policy
    no_small_values gives the same tables."""
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    assert ps.kernel_name(0, nreq, npts, has_verts=cells) == "fxk::tabulate_simplex_small"
    assert ps.kernel_name(1, nreq, npts, has_verts=cells) != "fxk::tabulate_simplex_small"
    pts, verts = batch(sd, nreq, npts, 3 * npts + nreq, cells)
    got = ps.tabulate_batch(0, pts, verts=verts).cpu().numpy()
    ref = oracle_tables(el, sd, 0, pts, verts, got.shape)
    assert rel(got, ref) <= TOL_VAL, rel(got, ref)
    kernel_policy("no_small_values")
    assert ps.kernel_name(0, nreq, npts, has_verts=cells) != "fxk::tabulate_simplex_small"
    assert rel(ps.tabulate_batch(0, pts, verts=verts).cpu().numpy(), ref) <= TOL_VAL


# (family, sd, degree, points) -> instance suffix of the request-per-workgroup kernel with the chain rule inside
WG_MIX = [("Lagrange", 3, 6, 122, "8>+mix"), ("DiscontinuousLagrange", 3, 5, 74, "6>+mix"), ("Lagrange", 3, 4, 97, "8>+mix"),
          ("Lagrange", 3, 3, 97, "8>+mix"), ("Nedelec", 3, 3, 74, "6>+mix"), ("Lagrange", 2, 6, 73, "6>+mix"), ("DiscontinuousLagrange", 2, 6, 79, "6>+mix"),
          ("Lagrange", 2, 6, 128, "8>+mix"), ("Lagrange", 3, 6, 100, "8>+mix"), ("Lagrange", 3, 5, 122, "8>+mix"),
          # 49..64 points: two requests per slab
          ("Nedelec", 3, 3, 57, "8>x2+mix"), ("Lagrange", 3, 6, 57, "8>x2+mix"), ("Lagrange", 3, 4, 64, "8>x2+mix"), ("Lagrange", 2, 6, 49, "8>x2+mix"),
          ("Lagrange", 2, 5, 55, "8>x2+mix"),
          # windows found by the off-default audits: vector-valued degree 3 at 13..15 points, odd tables without an 8-byte twin
          ("Nedelec", 3, 3, 14, "8>x9+mix"), ("RaviartThomas", 3, 3, 15, "8>x8+mix"), ("Lagrange", 2, 5, 19, "8>x6+mix"),
          ("Lagrange", 3, 4, 31, "8>x4+mix"), ("Lagrange", 2, 5, 21, "8>x6+mix"),
          # several small requests per slab (the default windows and policy wg_small), odd table sizes
          ("Lagrange", 2, 5, 25, "8>x5+mix"), ("DiscontinuousLagrange", 2, 5, 33, "8>x3+mix")]


@pytest.mark.parametrize("family,sd,degree,npts,suffix", WG_MIX, ids=[f"{m[0][:3]}{m[2]}-sd{m[1]}-{m[3]}pt" for m in WG_MIX])
@pytest.mark.parametrize("nreq", [1, 5, 333, 1030])
def test_chain_rule_inside_the_request_per_workgroup_kernel(family, sd, degree, npts, suffix, nreq, kernel_policy):
    """Per-request cells with gradients on the request-per-workgroup kernel (simplex_wg.hpp MIX 1): dof-major row tiles, the
    accumulators of the 1 + SD tables of a dof tile mixed with K = A0^-1 A_req of the column's request, table by table out under
    the next dof tile's MFMAs -- one pass (FIAT/expansions.py:411-447 through Jinv).  Cells of both orientations, batches smaller
    than the grid and of several groups per workgroup, a last group with missing requests, against the C oracle's recurrence ON the
    physical cells; the two-pass route (policy no_stacked_mix) gives the same tables.  (Degree-6 tetrahedra at 65..96 points keep the
    point-chunked chain-rule instances: their six-tile instance would spill.)"""
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    n = el.get_nodal_basis().get_embedded_degree()
    name = ps.kernel_name(1, nreq, npts, has_verts=True, instance=True)
    assert name == f"fxk::tabulate_simplex_wg<{sd},{n},{suffix}", name
    pts, verts = batch(sd, nreq, npts, 13 * npts + degree + nreq, True)
    got = ps.tabulate_batch(1, pts, verts=verts).cpu().numpy()
    ref = oracle_tables(el, sd, 1, pts, verts, got.shape)
    for t in range(got.shape[1]):
        assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (name, t, rel(got[:, t], ref[:, t]))
    kernel_policy("no_stacked_mix")
    assert "+mix" not in ps.kernel_name(1, nreq, npts, has_verts=True, instance=True)
    two = ps.tabulate_batch(1, pts, verts=verts).cpu().numpy()
    for t in range(got.shape[1]):
        assert rel(two[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER)


@pytest.mark.parametrize("family,sd,degree,npts,order,cells,suffix", [
    ("Lagrange", 2, 5, 19, 1, False, "8>x6"), ("Lagrange", 2, 5, 21, 2, False, None), ("Lagrange", 3, 4, 31, 0, False, "8>x4"),
    ("Lagrange", 3, 4, 31, 0, True, "8>x4"), ("Lagrange", 3, 4, 23, 0, False, None), ("Nedelec", 3, 3, 23, 1, True, None),
    ("Lagrange", 3, 5, 57, 0, False, "8>x2"), ("Lagrange", 3, 5, 57, 2, False, "8>x2"), ("Lagrange", 3, 4, 58, 0, False, None), ("Lagrange", 3, 4, 57, 0, False, "8>x2"),
    # vector-valued degree-3 tetrahedra of 180 rows a table at 17..24 points with cells, values only (windows the 0.8 GB audits
    # suggested beside it did not hold in 4 GB batches: not taken)
    ("RaviartThomas", 3, 3, 14, 2, False, None), ("Nedelec", 3, 3, 14, 1, False, None), ("BrezziDouglasMarini", 3, 3, 14, 0, True, None),
    ("Nedelec", 3, 3, 16, 1, False, None), ("BrezziDouglasMarini", 3, 3, 23, 0, True, "8>x5"), ("NedelecSecondKind", 3, 3, 23, 1, True, None),
    ("BrezziDouglasMarini", 3, 3, 23, 0, False, "8>x5"), ("NedelecSecondKind", 3, 3, 21, 0, False, "8>x6"), ("BrezziDouglasMarini", 3, 3, 23, 1, False, None),
    ("RaviartThomas", 3, 3, 23, 0, True, None), ("BrezziDouglasMarini", 3, 3, 57, 1, False, None), ("BrezziDouglasMarini", 3, 3, 57, 0, False, None),
    ("RaviartThomas", 3, 3, 57, 2, False, None),
    # degree >= 4 tetrahedra at 13..15 points (nine requests per slab), values of degree >= 5 at 25..32; P6 Hessians at 17..24 stay
    ("Lagrange", 3, 5, 14, 0, False, "8>x9"), ("Lagrange", 3, 6, 14, 2, True, "8>x9"), ("Lagrange", 3, 4, 14, 1, True, "8>x9+mix"),
    ("Lagrange", 3, 4, 14, 1, False, None), ("Lagrange", 3, 4, 15, 2, False, "8>x8"), ("Lagrange", 3, 5, 31, 0, True, "8>x4"),
    ("Lagrange", 3, 6, 23, 2, False, None), ("Lagrange", 3, 6, 23, 2, True, None), ("Lagrange", 3, 5, 13, 1, True, "8>x9+mix")])
def test_windows_of_grouped_requests(family, sd, degree, npts, order, cells, suffix, kernel_policy):
    """Where several requests per workgroup are the default below 65 points (round 4, sustained A/B in DESIGN.md 4.16): tables of an
    odd number of doubles at 17..48 points that only the point chunks held, degree >= 5 tetrahedra at 49..64 points -- and where they
    are not (shapes whose whole-request instance has an 8-byte twin, four-tile instances of lower degrees).  Tables against the oracle
    either way, and equal under policy no_wg."""
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    n = el.get_nodal_basis().get_embedded_degree()
    nreq = 517
    name = ps.kernel_name(order, nreq, npts, has_verts=cells, instance=True)
    if suffix is None:
        assert "simplex_wg" not in name, name
    else:
        assert name == f"fxk::tabulate_simplex_wg<{sd},{n},{suffix}", name
    pts, verts = batch(sd, nreq, npts, 5 * npts + order, cells)
    got = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    ref = oracle_tables(el, sd, order, pts, verts, got.shape)
    for t in range(got.shape[1]):
        assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (name, t)
    kernel_policy("no_wg")
    other = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    for t in range(got.shape[1]):
        assert rel(other[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER)


@pytest.mark.parametrize("family,sd,degree,npts,order,cells,kernel", [
    ("BrezziDouglasMarini", 2, 2, 6, 2, False, "small"), ("BrezziDouglasMarini", 2, 2, 12, 2, True, "small"), ("NedelecSecondKind", 2, 2, 6, 2, True, "small"),
    ("NedelecSecondKind", 2, 2, 3, 2, False, "small"), ("BrezziDouglasMarini", 3, 1, 4, 0, False, "small"), ("BrezziDouglasMarini", 3, 1, 4, 1, True, "small"),
    ("NedelecSecondKind", 3, 1, 11, 1, False, "small"), ("BrezziDouglasMarini", 3, 1, 4, 2, False, "kernel"), ("Nedelec", 3, 1, 4, 2, False, "small")])
@pytest.mark.parametrize("nreq", [1, 37, 2051])
def test_vector_valued_low_degree_requests_on_the_lane_local_kernel(family, sd, degree, npts, order, cells, kernel, nreq):
    """Second half of round 4: vector-valued degree-2 triangles with Hessians (24 rows, 6.9 KB a request at the 6-point rule: up to four
    requests share a wave now) and BDM1 / N2 tetrahedra of degree 1 with values / gradients (36 rows) moved from the generic kernel to
    the lane-local one; BDM1 with Hessians stays.  Tables against the oracle, odd batch sizes (a last wave with missing requests)."""
    import fiat_amd as fa
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    ps = el.device_polyset()
    assert ps.kernel_name(order, nreq, npts, has_verts=cells) == f"fxk::tabulate_simplex_{kernel}"
    pts, verts = batch(sd, nreq, npts, 7 * npts + order + nreq, cells)
    got = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    ref = oracle_tables(el, sd, order, pts, verts, got.shape)
    for t in range(got.shape[1]):
        assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (family, t, rel(got[:, t], ref[:, t]))


@pytest.mark.parametrize("family,sd,degree,npts,order", [
    ("Lagrange", 2, 1, 3, 2), ("Lagrange", 2, 1, 3, 1), ("Lagrange", 3, 1, 4, 2), ("DiscontinuousLagrange", 3, 0, 1, 2), ("DiscontinuousLagrange", 2, 0, 1, 1),
    ("Lagrange", 2, 2, 6, 2), ("Nedelec", 2, 1, 3, 1), ("RaviartThomas", 3, 1, 4, 0), ("BrezziDouglasMarini", 2, 1, 3, 2), ("Nedelec", 3, 1, 4, 1),
    ("Lagrange", 2, 2, 6, 0), ("Lagrange", 3, 2, 11, 1)])
@pytest.mark.parametrize("nreq", [1, 65, 4099])
def test_one_rule_in_many_cells_tiny_requests(family, sd, degree, npts, order, nreq, kernel_policy):
    """Second half of round 4: requests of <= 2 KB of tables with derivatives or a Piola map take the lane-local kernel on the ONE
    reference point set (SmallArgs::shared_pts) instead of the streaming kernels of shared_points.hpp.  Scalar elements: tables against
    the C oracle's recurrence on the physical cells; all: equal to the per-request-point path with the element's mapping and to the
    streaming kernels (policy no_small); the last two shapes stay on the streaming kernels (scalar values, > 2 KB)."""
    import fiat_amd as fa
    from oracle import fiat_oracle as fo
    el = getattr(fa, family)(fa.ufc_simplex(sd), degree)
    rng = np.random.default_rng(11 * npts + order + nreq)
    e = rng.exponential(size=(npts, sd + 1))
    bary = e / e.sum(axis=1, keepdims=True)
    ref_pts = bary @ fo.UFC_SIMPLEX[sd]
    A = np.eye(sd) + 0.2 * rng.standard_normal((nreq, sd, sd))
    A[::3, :, 0] *= -1.0
    verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[sd], A) + rng.standard_normal((nreq, 1, sd))
    pts = np.einsum("pv,rvd->rpd", bary, verts)
    got = el.tabulate_cells(order, ref_pts, verts).cpu().numpy()
    if el.mapping()[0] == "affine":
        ref = oracle_tables(el, sd, order, pts, verts, got.shape)
        for t in range(got.shape[1]):
            assert rel(got[:, t], ref[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (family, "oracle", t, rel(got[:, t], ref[:, t]))
    want = el.tabulate_batch(order, pts, verts=verts, pushforward=True).cpu().numpy()
    kernel_policy("no_small")
    old = el.tabulate_cells(order, ref_pts, verts).cpu().numpy()
    for t in range(got.shape[1]):
        assert rel(got[:, t], want[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (family, "per-request points", t)
        assert rel(got[:, t], old[:, t]) <= (TOL_VAL if t == 0 else TOL_DER), (family, "streaming kernels", t)
