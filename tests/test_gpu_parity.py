"""Parity of the HIP path (through the C ABI) against golden vectors generated
from the reference and against the CPU oracle on seeded batches.

Tolerances (BASELINE.json north_star): <= 1e-12 relative on values, <= 1e-10 on
derivatives, error norm  max|x - ref| / max(1, max|ref|)  per table."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu

TOL_VAL = 1e-12
TOL_DER = 1e-10


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from fiat_amd import runtime
    runtime.Context.get()
    return runtime


def table_errors(got, ref):
    """got/ref: (ntab, ...) -> list of per-table relative errors."""
    errs = []
    for g, r in zip(got, ref):
        errs.append(float(np.max(np.abs(g - r)) / max(1.0, np.max(np.abs(r)))))
    return errs


def assert_tables(got, ref, sd, what=""):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.all(np.isfinite(got)), what
    errs = table_errors(got, ref)
    assert errs[0] <= TOL_VAL, (what, "values", errs[0])
    if len(errs) > 1:
        assert max(errs[1:]) <= TOL_DER, (what, "derivatives", max(errs[1:]))


def rand_points(rng, sd, shape):
    e = rng.exponential(size=tuple(shape) + (sd + 1,))
    bary = e / e.sum(axis=-1, keepdims=True)
    return bary[..., 1:].copy()


# ---- a5-a7: raw expansion sets -------------------------------------------------------
@pytest.mark.parametrize("sd", [1, 2, 3])
@pytest.mark.parametrize("variant", [None, "bubble", "dual"])
def test_expansion_tables_vs_reference(rt, golden, sd, variant):
    g = golden("expansion")
    n_checked = 0
    for ci in (0, 1):
        verts = g[f"verts_sd{sd}_c{ci}"]
        pts = g[f"cpts_sd{sd}_c{ci}"]
        for n in (0, 1, 2, 3, 4, 6):
            for order in (0, 1, 2):
                key = f"exp_sd{sd}_c{ci}_{variant}_n{n}_o{order}"
                if key not in g:
                    continue
                ps = rt.SimplexPolySet(sd, n, variant=variant, verts=verts)
                out = ps.tabulate_batch(order, pts[None]).cpu().numpy()[0]
                assert_tables(out, g[key], sd, key)
                n_checked += 1
    assert n_checked >= 20


@pytest.mark.parametrize("sd", [1, 2, 3])
def test_expansion_physical_cell_both_ways(rt, golden, sd):
    """Cell given at element creation and cell given per request must agree with
    the reference built on that physical cell."""
    g = golden("expansion")
    verts, pts = g[f"verts_sd{sd}_phys"], g[f"cpts_sd{sd}_phys"]
    for variant, scale in ((None, None), ("bubble", 1)):
        for n in (2, 3):
            ref = g[f"exp_sd{sd}_phys_{variant}_n{n}_o2"]
            ps = rt.SimplexPolySet(sd, n, variant=variant, scale=scale, verts=verts)
            assert_tables(ps.tabulate_batch(2, pts[None]).cpu().numpy()[0], ref, sd, "fixed cell")
            ps2 = rt.SimplexPolySet(sd, n, variant=variant, scale=scale)
            got = ps2.tabulate_batch(2, pts[None], verts=verts[None]).cpu().numpy()[0]
            assert_tables(got, ref, sd, "per-request cell")


def test_bad_arguments(rt):
    with pytest.raises(ValueError):
        rt.SimplexPolySet(4, 1)
    with pytest.raises(ValueError):
        rt.SimplexPolySet(2, 0, variant="bubble")
    ps = rt.SimplexPolySet(2, 2)
    with pytest.raises(NotImplementedError):
        ps.tabulate_batch(9, np.zeros((1, 2, 2)))          # orders 3..8 run through differentiation matrices
    with pytest.raises(NotImplementedError):          # ... with per-request cells too (round 4: any order <= 8, tests/test_gpu_round4.py)
        ps.tabulate_batch(9, np.zeros((1, 2, 2)), verts=np.array([[[0.0, 0], [1, 0], [0, 1]]]))
    assert ps.tabulate_batch(5, np.zeros((1, 2, 2)) + 0.25, verts=np.array([[[0.0, 0], [1, 0], [0, 1]]])).shape[1] == 21
    with pytest.raises(ValueError):
        ps.tabulate_batch(1, np.zeros((1, 2, 3)))
    # empty batches are fine
    assert ps.tabulate_batch(1, np.zeros((0, 5, 2))).shape == (0, 3, 6, 5)
    assert ps.tabulate_batch(1, np.zeros((3, 0, 2))).shape == (3, 3, 6, 0)


# ---- C1: P1 triangle ---------------------------------------------------------------
def test_c1_p1_triangle(rt, golden):
    g = golden("elements")
    ps = rt.SimplexPolySet(2, 1, variant="bubble", scale=1, coeffs=g["c1_p1tri_coeffs"])
    out = ps.tabulate_batch(1, g["c1_p1tri_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["c1_p1tri_tab"], 2, "C1")


# ---- C2: P3 tetrahedron ------------------------------------------------------------
def test_c2_p3_tet_golden(rt, golden):
    g = golden("elements")
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=g["c2_p3tet_q6_coeffs"])
    out = ps.tabulate_batch(1, g["tet_q6_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["c2_p3tet_q6_tab"], 3, "C2 q6")
    out = ps.tabulate_batch(1, g["c2_p3tet_rand_pts"]).cpu().numpy()
    for o, r in zip(out, g["c2_p3tet_rand_tab"]):
        assert_tables(o, r, 3, "C2 random")
    out = ps.tabulate_batch(2, g["c2_p3tet_rand_pts"][:1]).cpu().numpy()[0]
    assert_tables(out, g["c2_p3tet_o2_tab"], 3, "C2 order 2")


@pytest.mark.parametrize("nreq", [1, 2, 3, 64, 1001])
def test_c2_batch_vs_oracle(rt, golden, nreq):
    """Ragged batch sizes (odd counts exercise the packing tail)."""
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    rng = np.random.default_rng(2)
    pts = rand_points(rng, 3, (nreq, 23))
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    out = ps.tabulate_batch(1, pts).cpu().numpy()
    idx = range(nreq) if nreq <= 64 else rng.choice(nreq, 48, replace=False)
    for r in idx:
        ref = fo.element_tabulate(fo.UFC_SIMPLEX[3], 3, co, 1, pts[r], 1, "bubble")
        ref = np.stack([ref[a] for a in fo.jet_indices(3, 1)])
        assert_tables(out[r], ref, 3, f"request {r}")


@pytest.mark.parametrize("nreq", [1, 2, 7, 515, 20001])
@pytest.mark.parametrize("npts", [21, 23, 24])
def test_c2_batch_physical_cells_vs_oracle(rt, golden, nreq, npts):
    """Per-request cell geometry (random affine images of the UFC tetrahedron) on the specialised
    kernel, odd and even batch sizes (two requests share a wave), every table of every request
    against the C restatement of the oracle."""
    from oracle import c_oracle
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    rng = np.random.default_rng(100 + nreq + npts)
    ref_pts = rand_points(rng, 3, (nreq, npts))
    A = np.eye(3) + 0.1 * rng.standard_normal((nreq, 3, 3))   # well-conditioned cells: the bound is on rounding, not on the map
    b = rng.standard_normal((nreq, 1, 3))
    verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[3], A) + b
    pts = np.einsum("rpd,red->rpe", ref_pts, A) + b          # points inside the physical cells
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    if 21 <= npts <= 24:
        assert ps.kernel_name(1, nreq, npts, has_verts=True) == "fxk::tabulate_simplex_pair"
    out = ps.tabulate_batch(1, pts, verts=verts).cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, pts, verts=verts, scale=1, variant="bubble")
    num = np.abs(out - ref).max(axis=(2, 3))
    den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
    err = (num / den).max(axis=0)
    assert err[0] <= TOL_VAL, err
    assert err[1:].max() <= TOL_DER, err


def test_c2_full_baseline_batch(rt, golden):
    """BASELINE config 2 at full size: 100 000 requests x 23 points, every table of every
    request against the C restatement of the oracle (pinned in tests/test_oracle_c.py)."""
    from oracle import c_oracle
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    rng = np.random.default_rng(2)
    pts = rand_points(rng, 3, (100000, 23))
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    out = ps.tabulate_batch(1, pts).cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, pts, scale=1, variant="bubble")
    num = np.abs(out - ref).max(axis=(2, 3))
    den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
    err = (num / den).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err
    # size-independent property: the Lagrange basis is a partition of unity, its gradient sums to zero
    assert np.abs(out[:, 0].sum(axis=1) - 1.0).max() < 1e-12
    assert np.abs(out[:, 1:].sum(axis=2)).max() < 1e-10


def test_c2_physical_cells(rt, golden):
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]   # affine-invariant
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    out = ps.tabulate_batch(1, g["c2_phys_pts"], verts=g["c2_phys_verts"]).cpu().numpy()
    for o, r in zip(out, g["c2_phys_tab"]):
        assert_tables(o, r, 3, "C2 physical")


@pytest.mark.parametrize("npts", [1, 3, 16, 17, 64, 65, 130])
def test_point_counts(rt, golden, npts):
    """Empty/ragged point counts: one point, tile boundaries, more than a wave."""
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    rng = np.random.default_rng(npts)
    pts = rand_points(rng, 3, (5, npts))
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    out = ps.tabulate_batch(1, pts).cpu().numpy()
    for r in range(5):
        ref = fo.element_tabulate(fo.UFC_SIMPLEX[3], 3, co, 1, pts[r], 1, "bubble")
        assert_tables(out[r], np.stack([ref[a] for a in fo.jet_indices(3, 1)]), 3, f"npts={npts}")


# ---- C3: N2 / RT2 (vector valued) ----------------------------------------------------
@pytest.mark.parametrize("name,n", [("n2", 2), ("rt2", 2), ("n1", 1), ("rt1", 1)])
def test_c3_hcurl_hdiv(rt, golden, name, n):
    g = golden("elements")
    tag = f"c3_{name}tet_q6" if n == 2 else f"c3_{name}tet"
    co = g[f"{tag}_coeffs"]
    ps = rt.SimplexPolySet(3, n, coeffs=co, value_shape=(3,))
    out = ps.tabulate_batch(1, g[f"{tag}_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g[f"{tag}_tab"], 3, tag)
    if n == 2:
        out = ps.tabulate_batch(1, g["c2_p3tet_rand_pts"][:2]).cpu().numpy()
        for o, r in zip(out, g[f"c3_{name}tet_rand_tab"]):
            assert_tables(o, r, 3, tag + " random")


def test_c3_mixed_batch(rt, golden):
    """The C3 workload: N2 and RT2 requests interleaved = two launches on
    disjoint halves of the batch."""
    g = golden("elements")
    rng = np.random.default_rng(3)
    pts = rand_points(rng, 3, (200, 23))
    for name in ("n2", "rt2"):
        co = g[f"c3_{name}tet_q6_coeffs"]
        ps = rt.SimplexPolySet(3, 2, coeffs=co, value_shape=(3,))
        sub = pts[0::2] if name == "n2" else pts[1::2]
        out = ps.tabulate_batch(1, sub).cpu().numpy()
        for r in (0, 37, 99):
            ref = fo.element_tabulate(fo.UFC_SIMPLEX[3], 2, co, 1, sub[r])
            assert_tables(out[r], np.stack([ref[a] for a in fo.jet_indices(3, 1)]), 3, name)


# ---- C4: DG P6 tet, Hessians ---------------------------------------------------------
def test_c4_dg6(rt, golden):
    g = golden("elements")
    co = g["c4_dg6tet_q6_coeffs"]
    ps = rt.SimplexPolySet(3, 6, coeffs=co)
    out = ps.tabulate_batch(2, g["tet_q6_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["c4_dg6tet_q6_tab"], 3, "C4 q6")
    out = ps.tabulate_batch(2, g["c2_p3tet_rand_pts"][1:2]).cpu().numpy()[0]
    assert_tables(out, g["c4_dg6tet_rand_tab"], 3, "C4 random")
    out = ps.tabulate_batch(2, g["tet_q12_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["c4_dg6tet_q12_tab"], 3, "C4 q12 (122 points)")


@pytest.mark.parametrize("sd", [2, 3])
@pytest.mark.parametrize("deg", [1, 2, 3, 4])
def test_lagrange_dg_families(rt, golden, sd, deg):
    g = golden("elements")
    for fam, variant, scale in (("lag", "bubble", 1), ("dg", None, None)):
        tag = f"{fam}_sd{sd}_p{deg}"
        ps = rt.SimplexPolySet(sd, deg, variant=variant, scale=scale, coeffs=g[f"{tag}_coeffs"])
        out = ps.tabulate_batch(2, g[f"{tag}_pts"][None]).cpu().numpy()[0]
        assert_tables(out, g[f"{tag}_tab"], sd, tag)


# ---- a11/a12: Vandermonde assembly and solve on the device -------------------------------
@pytest.mark.parametrize("tag,sd,n,variant,scale", [
    ("c2_p3tet_q6", 3, 3, "bubble", 1), ("c4_dg6tet_q6", 3, 6, None, None),
    ("lag_sd2_p4", 2, 4, "bubble", 1), ("dg_sd3_p2", 3, 2, None, None)])
def test_vandermonde_point_evaluation(rt, golden, tag, sd, n, variant, scale):
    g = golden("elements")
    verts = fo.UFC_SIMPLEX[sd]
    nodes, _ = (fo.lagrange_nodes if variant == "bubble" else fo.broken_lagrange_nodes)(verts, n)
    nodes = np.array(nodes)
    es = rt.SimplexPolySet(sd, n, variant=variant, scale=scale)
    # expansion values at the nodes: (nexp, nnodes); Riesz matrix rows = functionals
    ev = es.tabulate_batch(0, nodes[None])[0, 0]
    import torch
    wts = torch.eye(len(nodes), dtype=torch.float64, device=ev.device)
    R = rt.riesz_assemble(wts, ev)
    B = torch.eye(len(nodes), dtype=torch.float64, device=ev.device)
    X, V = rt.vandermonde_solve_batch(R, B, return_V=True)
    V = V.cpu().numpy()[0]
    X = X.cpu().numpy()[0]
    assert np.max(np.abs(V - g[f"{tag}_V"])) / np.max(np.abs(g[f"{tag}_V"])) < 1e-13
    ref = g[f"{tag}_coeffs"]
    assert np.max(np.abs(X - ref)) / np.max(np.abs(ref)) < 1e-11


def test_vandermonde_singular_raises(rt):
    A = np.ones((1, 4, 4))
    B = np.eye(4)[None]
    with pytest.raises(np.linalg.LinAlgError):
        rt.vandermonde_solve_batch(A, B)


def test_vandermonde_batch_random(rt):
    rng = np.random.default_rng(7)
    A = rng.normal(size=(33, 20, 30))
    B = rng.normal(size=(33, 20, 30))
    X = rt.vandermonde_solve_batch(A, B).cpu().numpy()
    for s in (0, 16, 32):
        V = A[s] @ B[s].T
        ref = np.linalg.solve(V.T, B[s])
        assert np.max(np.abs(X[s] - ref)) / np.max(np.abs(ref)) < 1e-9 * np.linalg.cond(V)


# ---- a8 / a14: 1-D Lagrange and tensor products -------------------------------------------
def test_line_lagrange(rt, golden):
    g = golden("lagrange_line")
    L = rt.LineLagrange(g["nodes"])
    out = L.tabulate_batch(2, g["pts"].reshape(1, -1)).cpu().numpy()[0]
    assert_tables(out, g["tab"], 1, "line")
    assert np.array_equal(out[0][:, -1], np.eye(5)[:, 4])   # node hit -> exact delta


def test_c5_hex(rt, golden):
    g = golden("tensor_product")
    L = rt.LineLagrange(g["p4_nodes"])
    out = rt.tensor_tabulate_batch([L, L, L], 1, g["hex_rand_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["hex_rand_tab"], 3, "hex random points")
    out = rt.tensor_tabulate_batch([L, L, L], 1, g["hex_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["hex_tab"], 3, "hex grid points")


def test_c5_hex_grid_mode_matches_point_mode(rt):
    rng = np.random.default_rng(5)
    nodes = np.array([0.0, 1.0, 0.25, 0.5, 0.75])
    L = rt.LineLagrange(nodes)
    grid = np.sort(rng.uniform(0, 1, size=(7, 3, 5)), axis=2)
    pts = np.stack([np.array([[x, y, z] for x in g[0] for y in g[1] for z in g[2]]) for g in grid])
    a = rt.tensor_tabulate_batch([L, L, L], 1, grid, grid=True).cpu().numpy()
    b = rt.tensor_tabulate_batch([L, L, L], 1, pts).cpu().numpy()
    assert a.shape == b.shape == (7, 4, 125, 125)
    assert np.array_equal(a, b)
    ref = fo.hex_lagrange_tabulate(nodes, 1, pts[3])
    assert_tables(a[3], np.stack([ref[al] for al in fo.jet_indices(3, 1)]), 3, "hex vs oracle")


def test_quad_mixed(rt, golden):
    g = golden("tensor_product")
    La = rt.LineLagrange(g["p4_nodes"])
    Lb = rt.LineLagrange(np.array([0.0, 1.0, 0.5]))
    out = rt.tensor_tabulate_batch([La, Lb], 2, g["quad_pts"][None]).cpu().numpy()[0]
    assert_tables(out, g["quad_tab"], 2, "quad")


# ---- kernel selection: the benchmark shapes must run on their specialised kernels -----
def test_kernel_selection(rt, golden):
    g = golden("elements")
    p3 = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=g["c2_p3tet_q6_coeffs"])
    assert p3.kernel_name(1, 1000, 23) == "fxk::tabulate_simplex_pair"
    assert p3.kernel_name(1, 1000, 23, has_verts=True) == "fxk::tabulate_simplex_pair"
    assert p3.kernel_name(1, 1000, 40, has_verts=True) == "fxk::tabulate_simplex_pair"   # one request per wave, 10 column tiles
    assert p3.kernel_name(1, 1000, 40) == "fxk::tabulate_simplex_stacked"    # own cell, off the 21..24-point shape: the paired entry yields
    assert p3.kernel_name(1, 1000, 10) == "fxk::tabulate_simplex_pair"       # ... unless no stacked instance holds the request
    assert p3.kernel_name(1, 1000, 50) == "fxk::tabulate_simplex_stacked"    # 49..64 points: four column tiles
    assert p3.kernel_name(1, 1000, 100) == "fxk::tabulate_simplex_wg"        # 97..128 points: a request per workgroup (round 4; were point chunks)
    assert p3.kernel_name(1, 1000, 70) == "fxk::tabulate_simplex_stacked"    # 65..96 points with a short K loop: point-chunked units
    assert p3.kernel_name(1, 1000, 130) == "fxk::tabulate_simplex_stacked"   # point-chunked units
    assert p3.kernel_name(1, 1000, 7) == "fxk::tabulate_simplex_kernel"      # fewer points than any registered tiling
    assert p3.kernel_name(2, 1000, 23) == "fxk::tabulate_simplex_stacked"    # Hessians: 200 stacked rows
    assert p3.kernel_name(2, 1000, 23, has_verts=True) == "fxk::tabulate_simplex_stacked"   # + table-mixing pass
    dg6 = rt.SimplexPolySet(3, 6, coeffs=g["c4_dg6tet_q6_coeffs"])
    assert dg6.kernel_name(2, 1000, 23) == "fxk::tabulate_simplex_stacked"          # requests on the element's cell
    assert dg6.kernel_name(2, 1000, 23, has_verts=True) == "fxk::tabulate_simplex_stacked"  # per-request cells: + table-mixing pass
    assert dg6.kernel_name(2, 1000, 122) == "fxk::tabulate_simplex_wg"              # a request per workgroup (C4 stress variant; round 3: point chunks)
    assert dg6.kernel_name(2, 1000, 200) == "fxk::tabulate_simplex_stacked"         # point-chunked units


def test_concurrent_streams(rt, golden):
    """Launches of the dynamically scheduled kernel in flight on two streams at once: every launch
    gets its own work counter (the counter resets itself when the launch ends)."""
    import torch
    from oracle import c_oracle
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    rng = np.random.default_rng(11)
    nreq = 6001
    pts = [rand_points(rng, 3, (nreq, 23)) for _ in range(2)]
    dpts = [torch.as_tensor(p).cuda() for p in pts]
    outs = [torch.empty(ps.out_shape(1, nreq, 23), dtype=torch.float64, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rep in range(20):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                ps.tabulate_batch(1, dpts[k], out=outs[k], stream=streams[k])
    torch.cuda.synchronize()
    for k in range(2):
        ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, pts[k], scale=1, variant="bubble")
        got = outs[k].cpu().numpy()
        num = np.abs(got - ref).max(axis=(2, 3))
        den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
        err = (num / den).max(axis=0)
        assert err[0] <= TOL_VAL and err[1:].max() <= TOL_DER, (k, err)


@pytest.mark.parametrize("npts", [9, 11, 12, 13, 14, 16, 17, 19, 20, 25, 27, 29, 31, 32, 33, 36, 40, 41, 44, 46, 48])
@pytest.mark.parametrize("nreq,cells", [(1, False), (2, True), (515, False), (1030, True)])
def test_p3_tet_paired_kernel_point_counts(rt, golden, kernel_policy, npts, nreq, cells):
    """The paired kernel is instantiated for 3..6 column tiles (two requests per wave: P3 tetrahedra
    with 9..24 points) and for 8 / 10 / 12 tiles with one request per wave (25..48 points).  On the element's own cell
    these entries yield to the stacked kernel by default (plan_launch, fixed_yields): policy "no_stacked" keeps them."""
    from oracle import c_oracle
    co = golden("elements")["c2_p3tet_q6_coeffs"]
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    rng = np.random.default_rng(3000 + 10 * npts + nreq)
    pts = rand_points(rng, 3, (nreq, npts))
    verts = None
    if cells:
        A = np.eye(3) + 0.1 * rng.standard_normal((nreq, 3, 3))
        b = rng.standard_normal((nreq, 1, 3))
        verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[3], A) + b
        pts = np.einsum("rpd,red->rpe", pts, A) + b
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, pts, verts=verts, scale=1, variant="bubble")

    def check(route):
        out = ps.tabulate_batch(1, pts, verts=verts).cpu().numpy()
        num = np.abs(out - ref).max(axis=(2, 3))
        den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
        err = (num / den).max(axis=0)
        assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), (route, err)

    if not cells:
        if npts >= 12:
            assert ps.kernel_name(1, nreq, npts) == "fxk::tabulate_simplex_stacked"
            check("default route: stacked")        # the route a user gets, against the oracle, before the policy switch
        kernel_policy("no_stacked")
    assert ps.kernel_name(1, nreq, npts, has_verts=cells) == "fxk::tabulate_simplex_pair"
    check("pair")


@pytest.mark.parametrize("fam,deg", [("Lagrange", 4), ("RaviartThomas", 2), ("DiscontinuousLagrange", 4)])
@pytest.mark.parametrize("npts", [21, 23, 24])
@pytest.mark.parametrize("nreq,cells", [(1, False), (3, True), (2049, False), (1500, True)])
def test_one_request_per_wave_instances(rt, kernel_policy, fam, deg, npts, nreq, cells):
    """P4 / RT2 tetrahedra: the K-streamed kernel with one request per wave and a half-request image (P4 on the element's
    own cell: behind policy "no_stacked", the entry yields to the stacked kernel by default)."""
    import fiat_amd
    from oracle import c_oracle
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(3), deg)
    ps = el.device_polyset()
    rng = np.random.default_rng(7000 + 10 * npts + nreq + deg)
    pts = rand_points(rng, 3, (nreq, npts))
    verts = None
    if cells:
        A = np.eye(3) + 0.1 * rng.standard_normal((nreq, 3, 3))
        b = rng.standard_normal((nreq, 1, 3))
        verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[3], A) + b
        pts = np.einsum("rpd,red->rpe", pts, A) + b
    n = deg if fam != "RaviartThomas" else 2
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], n, el.get_coeffs(), 1, pts, verts=verts, scale=el._expansion_scale,
                                  variant=el._expansion_variant)

    def check(route):
        out = ps.tabulate_batch(1, pts, verts=verts).cpu().numpy()
        r = ref.reshape(out.shape)
        axes = tuple(range(2, out.ndim))
        num = np.abs(out - r).max(axis=axes)
        den = np.maximum(1.0, np.abs(r).max(axis=axes))
        err = (num / den).max(axis=0)
        assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), (route, err)

    if not cells and deg == 4:
        if (35 * 4 * npts) % 2 == 0:
            assert ps.kernel_name(1, nreq, npts) == "fxk::tabulate_simplex_stacked"
            check("default route: stacked")
        kernel_policy("no_stacked")
    assert ps.kernel_name(1, nreq, npts, has_verts=cells) == "fxk::tabulate_simplex_pair"
    check("pair")


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("npts,nreq", [(13, 8), (16, 100), (17, 5), (23, 1), (23, 2050), (24, 333), (25, 64), (32, 129), (33, 7),
                                       (40, 100), (48, 65), (49, 3), (64, 130)])
def test_stacked_matrix_kernel(rt, golden, order, npts, nreq, kernel_policy):
    """Degree-6 tetrahedron on the element's own cell: all derivative tables as rows of ONE stacked matrix
    [C; C D^alpha] (simplex_stacked.hpp), every (column tiles, requests per group) instance, odd batch sizes
    (a last group with a missing request), against the C oracle's recurrence derivatives; and a physical
    element cell (the derivative matrices are taken on the element's cell, not the UFC one)."""
    from oracle import c_oracle
    g = golden("elements")
    co = g["c4_dg6tet_q6_coeffs"]
    ps = rt.SimplexPolySet(3, 6, coeffs=co)
    # (round 4: the default route of 49..128 points -- and of 33..48 points with Hessians -- is the request-per-workgroup kernel,
    # tests/test_gpu_round4.py)
    # (... and of 13..15 points, nine requests per slab)
    if npts > 48 or (order == 2 and npts >= 33) or 13 <= npts <= 15:
        assert ps.kernel_name(order, nreq, npts) == "fxk::tabulate_simplex_wg"
        kernel_policy("no_wg")
    assert ps.kernel_name(order, nreq, npts) == "fxk::tabulate_simplex_stacked"
    rng = np.random.default_rng(100 * npts + nreq + order)
    pts = rand_points(rng, 3, (nreq, npts))
    out = ps.tabulate_batch(order, pts).cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 6, co, order, pts).reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err


def test_stacked_matrix_kernel_on_a_physical_element_cell(rt):
    import fiat_amd
    from oracle import c_oracle
    rng = np.random.default_rng(99)
    A = np.eye(3) + 0.2 * rng.standard_normal((3, 3))
    verts = fo.UFC_SIMPLEX[3] @ A.T + rng.standard_normal(3)
    el = fiat_amd.DiscontinuousLagrange(fiat_amd.physical_simplex(verts), 6)
    ps = el.device_polyset()
    nreq, npts = 37, 23
    assert ps.kernel_name(2, nreq, npts) == "fxk::tabulate_simplex_stacked"
    e = rng.exponential(size=(nreq, npts, 4))
    pts = (e / e.sum(axis=-1, keepdims=True)) @ verts
    out = ps.tabulate_batch(2, pts).cpu().numpy()
    ref = c_oracle.tabulate_batch(verts, 6, el.get_coeffs(), 2, pts, scale=el._expansion_scale,
                                  variant=el._expansion_variant).reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err


@pytest.mark.parametrize("fam,deg,order,npts", [
    ("Lagrange", 3, 2, 23), ("Lagrange", 4, 2, 23), ("Lagrange", 5, 1, 23), ("Lagrange", 5, 2, 30), ("Lagrange", 6, 2, 23),
    ("DiscontinuousLagrange", 5, 1, 40), ("DiscontinuousLagrange", 4, 2, 19), ("Nedelec", 3, 1, 23), ("Nedelec", 4, 1, 23),
    ("Nedelec", 4, 2, 17), ("RaviartThomas", 3, 1, 28), ("BrezziDouglasMarini", 3, 1, 23), ("NedelecSecondKind", 3, 2, 23),
    ("Lagrange", 5, 0, 23), ("Lagrange", 4, 1, 30), ("Lagrange", 3, 1, 55), ("DiscontinuousLagrange", 6, 0, 14)])
def test_stacked_matrix_kernel_families(fam, deg, order, npts, kernel_policy):
    """The stacked-matrix kernel across expansion degrees 3-6, bubble (Lagrange: the C0 transform is folded
    into the coefficients, the derivative matrices are those of the raw hierarchy) and orthonormal variants,
    scalar and vector-valued elements, against the C oracle's recurrence derivatives."""
    import fiat_amd
    from oracle import c_oracle
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(3), deg)
    ps = el.device_polyset()
    nreq = 211
    kernel_policy("no_wg")   # (the request-per-workgroup kernel has taken some of these shapes since: tests/test_gpu_round4.py)
    assert ps.kernel_name(order, nreq, npts) == "fxk::tabulate_simplex_stacked"
    rng = np.random.default_rng(17 * deg + npts + order)
    pts = rand_points(rng, 3, (nreq, npts))
    out = ps.tabulate_batch(order, pts).cpu().numpy()
    n = el.get_nodal_basis().get_embedded_degree()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], n, el.get_coeffs(), order, pts, scale=el._expansion_scale,
                                  variant=el._expansion_variant).reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err


@pytest.mark.parametrize("npts,nreq", [(23, 4001), (17, 3), (28, 500), (40, 77)])
def test_stacked_matrix_kernel_small_shape_ab(rt, golden, kernel_policy, npts, nreq):
    """The register-resident instances of the stacked-matrix kernel (P3 tetrahedron, values + gradient: the A/B
    partner of the paired kernel on the benchmark shape, policy "stacked_small") against the C oracle."""
    from oracle import c_oracle
    kernel_policy("stacked_small")
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    ps = rt.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=co)
    assert ps.kernel_name(1, nreq, npts) == "fxk::tabulate_simplex_stacked"
    rng = np.random.default_rng(npts + nreq)
    pts = rand_points(rng, 3, (nreq, npts))
    out = ps.tabulate_batch(1, pts).cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, pts, scale=1.0, variant="bubble").reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err
    kernel_policy()
    assert ps.kernel_name(1, nreq, npts) == ("fxk::tabulate_simplex_pair" if 21 <= npts <= 24 else "fxk::tabulate_simplex_stacked")


@pytest.mark.parametrize("fam,deg,order,npts", [("Lagrange", 5, 1, 16), ("Lagrange", 5, 2, 23), ("Lagrange", 6, 1, 23),
                                                ("DiscontinuousLagrange", 6, 2, 30), ("Nedelec", 5, 1, 23),
                                                ("RaviartThomas", 6, 2, 52), ("Lagrange", 6, 2, 13)])
def test_stacked_matrix_kernel_triangles(fam, deg, order, npts):
    import fiat_amd
    from oracle import c_oracle
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(2), deg)
    ps = el.device_polyset()
    nreq = 301
    assert ps.kernel_name(order, nreq, npts) == "fxk::tabulate_simplex_stacked"
    rng = np.random.default_rng(23 * deg + npts + order)
    pts = rand_points(rng, 2, (nreq, npts))
    out = ps.tabulate_batch(order, pts).cpu().numpy()
    n = el.get_nodal_basis().get_embedded_degree()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[2], n, el.get_coeffs(), order, pts, scale=el._expansion_scale,
                                  variant=el._expansion_variant).reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err


@pytest.mark.parametrize("fam,sd,deg,order,npts", [("DiscontinuousLagrange", 3, 6, 2, 23), ("DiscontinuousLagrange", 3, 6, 1, 40),
                                                   ("Lagrange", 3, 4, 2, 23), ("Lagrange", 3, 5, 1, 30), ("Nedelec", 3, 3, 1, 23),
                                                   ("Lagrange", 2, 6, 2, 16), ("RaviartThomas", 2, 5, 1, 23),
                                                   ("Lagrange", 3, 5, 0, 23), ("Nedelec", 3, 4, 1, 24),
                                                   ("BrezziDouglasMarini", 3, 3, 1, 23), ("DiscontinuousLagrange", 3, 4, 1, 44),
                                                   ("DiscontinuousLagrange", 3, 6, 1, 30), ("Lagrange", 2, 6, 1, 23),
                                                   ("DiscontinuousLagrange", 2, 5, 1, 30), ("Lagrange", 2, 5, 1, 44),
                                                   ("Nedelec", 2, 6, 1, 22)])
@pytest.mark.parametrize("mix", ["1", "0"])
def test_stacked_matrix_kernel_with_per_request_cells(kernel_policy, mix, fam, sd, deg, order, npts):
    """Per-request cells on the stacked-matrix kernel: points mapped through the request's cell in the kernel,
    chain rule across the derivative tables by the in-place mixing pass (table_mix_kernel), against the C
    oracle's recurrence on the physical cells; one negatively oriented cell."""
    import fiat_amd
    from oracle import c_oracle
    # order 1: chain rule inside the kernel (MIXT instances) or, policy "no_stacked_mix", the mixing pass
    # (policy no_wg: the request-per-workgroup kernel has taken some of these shapes since -- its own tests are tests/test_gpu_round4.py)
    kernel_policy(*(["no_stacked_mix"] if mix == "0" else ["no_wg"]))
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
    ps = el.device_polyset()
    nreq = 131 if mix == "0" else 4133   # (several groups per wave for the in-kernel variant)
    assert ps.kernel_name(order, nreq, npts, has_verts=True) == "fxk::tabulate_simplex_stacked"
    rng = np.random.default_rng(31 * deg + npts + order + sd)
    A = np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))
    A[7, :, 0] *= -1.0
    verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[sd], A) + rng.standard_normal((nreq, 1, sd))
    e = rng.exponential(size=(nreq, npts, sd + 1))
    pts = np.einsum("rpv,rvd->rpd", e / e.sum(axis=-1, keepdims=True), verts)
    out = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    n = el.get_nodal_basis().get_embedded_degree()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[sd], n, el.get_coeffs(), order, pts, verts=verts, scale=el._expansion_scale,
                                  variant=el._expansion_variant).reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err


@pytest.mark.parametrize("fam,sd,deg,order,npts,cells", [
    ("DiscontinuousLagrange", 3, 6, 2, 122, False), ("DiscontinuousLagrange", 3, 6, 1, 65, True), ("Lagrange", 3, 3, 1, 70, False),
    ("Lagrange", 3, 5, 1, 97, False), ("Lagrange", 2, 6, 2, 73, True), ("DiscontinuousLagrange", 2, 5, 1, 15, False),
    ("Nedelec", 3, 3, 1, 49, False), ("Lagrange", 3, 4, 2, 200, False), ("RaviartThomas", 3, 3, 1, 23, True)])
def test_stacked_matrix_kernel_point_chunks(fam, sd, deg, order, npts, cells, kernel_policy):
    """Point-chunked units of the stacked-matrix kernel (16 CT points of one request per unit, 8-byte row stores): more
    than 64 points per request (the C4 stress variant: 122 points), a last chunk of any size, and odd table sizes
    that the 16-byte whole-request instances cannot take; against the C oracle."""
    import fiat_amd
    from oracle import c_oracle
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
    ps = el.device_polyset()
    nreq = 57
    kernel_policy("no_wg")   # (round 4: on the element's own cell rules of 49..128 points take the request-per-workgroup kernel by default)
    assert ps.kernel_name(order, nreq, npts, has_verts=cells) == "fxk::tabulate_simplex_stacked"
    rng = np.random.default_rng(7 * deg + npts + order + sd)
    pts = rand_points(rng, sd, (nreq, npts))
    verts = None
    if cells:
        A = np.eye(sd) + 0.1 * rng.standard_normal((nreq, sd, sd))
        b = rng.standard_normal((nreq, 1, sd))
        verts = np.einsum("vd,red->rve", fo.UFC_SIMPLEX[sd], A) + b
        pts = np.einsum("rpd,red->rpe", pts, A) + b
    out = ps.tabulate_batch(order, pts, verts=verts).cpu().numpy()
    n = el.get_nodal_basis().get_embedded_degree()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[sd], n, el.get_coeffs(), order, pts, verts=verts, scale=el._expansion_scale,
                                  variant=el._expansion_variant).reshape(out.shape)
    axes = tuple(range(2, out.ndim))
    err = (np.abs(out - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= TOL_VAL and (len(err) == 1 or err[1:].max() <= TOL_DER), err
