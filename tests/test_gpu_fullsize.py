"""BASELINE.json configs at their full per-GPU sizes, through size-independent properties
(evaluated on the device: the tables of C4 / C5 are 19 GB / 100 GB) and, where the CPU oracle
finishes in seconds, against the oracle on every request.

  C3  N2 + RT2 tetrahedra, order 1, 25 000 + 25 000 requests: all tables vs the C oracle
  C4  DG P6 tetrahedron, order 2, 125 000 requests (1 M / 8 GPUs): partition of unity,
      derivatives of the constant vanish, a sample of requests vs the C oracle
  C5  P4 x P4 x P4 hexahedron, order 1, 200 000 requests on 5^3 grids: partition of unity,
      sum of gradients zero, a sample vs the oracle"""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("name", ["n2", "rt2"])
def test_c3_full_size(rt, golden, name):
    from oracle import c_oracle
    co = golden("elements")[f"c3_{name}tet_q6_coeffs"]
    rng = np.random.default_rng(3 if name == "n2" else 33)
    pts = simplex_points(rng, 3, (25000, 23))
    ps = rt.SimplexPolySet(3, 2, coeffs=co, value_shape=(3,))
    out = ps.tabulate_batch(1, pts).cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 2, co, 1, pts).reshape(out.shape)
    num = np.abs(out - ref).max(axis=(2, 3, 4))
    den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3, 4)))
    err = (num / den).max(axis=0)
    assert err[0] <= 1e-12 and err[1:].max() <= 1e-10, err


def test_c4_full_size(rt, golden):
    import torch
    from oracle import c_oracle
    co = golden("elements")["c4_dg6tet_q6_coeffs"]
    nreq = 125000
    rng = np.random.default_rng(4)
    pts = simplex_points(rng, 3, (nreq, 23))
    ps = rt.SimplexPolySet(3, 6, coeffs=co)
    out = ps.tabulate_batch(2, pts)                       # (nreq, 10, 84, 23): 19.3 GB, stays on the device
    assert ps.kernel_name(2, nreq, 23) == "fxk::tabulate_simplex_stacked"
    s = out.sum(dim=2)                                    # sum over the basis: 1 for the values, 0 for every derivative
    assert float((s[:, 0] - 1.0).abs().max()) <= 1e-11
    assert float(s[:, 1:4].abs().max()) <= 1e-9          # gradients (entries up to ~1e2)
    assert float(s[:, 4:].abs().max()) <= 1e-7           # Hessians (entries up to ~1e4)
    assert bool(torch.isfinite(out).all())
    idx = rng.choice(nreq, 400, replace=False)
    got = out[torch.as_tensor(idx).cuda()].cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 6, co, 2, pts[idx]).reshape(got.shape)
    num = np.abs(got - ref).max(axis=(2, 3))
    den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
    err = (num / den).max(axis=0)
    assert err[0] <= 1e-12 and err[1:].max() <= 1e-10, err


def test_c5_full_size(rt, golden):
    import torch
    nodes = golden("tensor_product")["p4_nodes"]
    L = rt.LineLagrange(nodes)
    nreq = 200000
    rng = np.random.default_rng(5)
    grid = np.sort(rng.uniform(0, 1, size=(nreq, 3, 5)), axis=2)
    out = rt.tensor_tabulate_batch([L, L, L], 1, grid, grid=True)   # (nreq, 4, 125, 125): 100 GB on the device
    assert tuple(out.shape) == (nreq, 4, 125, 125)
    worst = [0.0, 0.0]
    for lo in range(0, nreq, 20000):                                 # reductions in slices (temporaries stay small)
        s = out[lo:lo + 20000].sum(dim=2)
        worst[0] = max(worst[0], float((s[:, 0] - 1.0).abs().max()))
        worst[1] = max(worst[1], float(s[:, 1:].abs().max()))
    assert worst[0] <= 1e-12 and worst[1] <= 1e-10, worst
    for r in (0, 99999, 199999):
        g = grid[r]
        pts = np.array([[x, y, z] for x in g[0] for y in g[1] for z in g[2]])
        ref = fo.hex_lagrange_tabulate(np.asarray(nodes), 1, pts)
        got = out[r].cpu().numpy()
        for t, al in enumerate(fo.jet_indices(3, 1)):
            assert np.abs(got[t] - ref[al]).max() <= (1e-12 if t == 0 else 1e-10) * max(1.0, np.abs(ref[al]).max())
    del out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("family,degree,order,nreq,key", [("Nedelec", 2, 1, 25000, "c3_n2tet_q6"), ("RaviartThomas", 2, 1, 25000, "c3_rt2tet_q6"),
                                                          ("DiscontinuousLagrange", 6, 2, 125000, "c4_dg6tet_q6")])
def test_full_size_with_device_constructed_elements(rt, golden, family, degree, order, nreq, key):
    """The same full-size batches through the façade element whose nodal coefficients were BUILT on the device
    (Riesz assembly + Vandermonde solve), not injected from the reference: coefficients equal the reference's, a sample
    of requests equals the C oracle evaluated with the reference's coefficients, every request is finite and -- DG --
    sums to one / zero over the basis."""
    import fiat_amd
    import torch
    from oracle import c_oracle
    el = getattr(fiat_amd, family)(fiat_amd.ufc_simplex(3), degree)
    ref_co = golden("elements")[key + "_coeffs"]
    assert np.abs(el.get_coeffs() - ref_co).max() <= 1e-11 * max(1.0, np.abs(ref_co).max())
    rng = np.random.default_rng(17 + degree + order)
    pts = simplex_points(rng, 3, (nreq, 23))
    out = el.tabulate_batch(order, pts)
    assert bool(torch.isfinite(out).all())
    if family == "DiscontinuousLagrange":
        s = out.sum(dim=2)
        assert float((s[:, 0] - 1.0).abs().max()) <= 1e-11 and float(s[:, 1:4].abs().max()) <= 1e-9
    idx = rng.choice(nreq, 300, replace=False)
    got = out[torch.as_tensor(idx).cuda()].cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], degree, ref_co, order, pts[idx]).reshape(got.shape)
    axes = tuple(range(2, got.ndim))
    err = (np.abs(got - ref).max(axis=axes) / np.maximum(1.0, np.abs(ref).max(axis=axes))).max(axis=0)
    assert err[0] <= 1e-12 and err[1:].max() <= 1e-10, err
    del out
    torch.cuda.empty_cache()


def test_c4_stress_122_points_full_size(rt, golden):
    """The C4 stress variant of bench.py --workload dg6tet122 (8000 requests of the 122-point rule, order 2, 6.6 GB of tables) on the
    request-per-workgroup kernel: partition of unity, vanishing derivative sums, a sample against the C oracle."""
    import torch
    from oracle import c_oracle
    co = golden("elements")["c4_dg6tet_q6_coeffs"]
    nreq, npts = 8000, 122
    rng = np.random.default_rng(122)
    pts = simplex_points(rng, 3, (nreq, npts))
    ps = rt.SimplexPolySet(3, 6, coeffs=co)
    assert ps.kernel_name(2, nreq, npts) == "fxk::tabulate_simplex_wg"
    out = ps.tabulate_batch(2, pts)
    s = out.sum(dim=2)
    assert float((s[:, 0] - 1.0).abs().max()) <= 1e-11
    assert float(s[:, 1:4].abs().max()) <= 1e-9
    assert float(s[:, 4:].abs().max()) <= 1e-7
    assert bool(torch.isfinite(out).all())
    idx = rng.choice(nreq, 60, replace=False)
    got = out[torch.as_tensor(idx).cuda()].cpu().numpy()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 6, co, 2, pts[idx]).reshape(got.shape)
    err = (np.abs(got - ref).max(axis=(2, 3)) / np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))).max(axis=0)
    assert err[0] <= 1e-12 and err[1:].max() <= 1e-10, err


@pytest.mark.parametrize("npts,nreq", [(74, 60001), (75, 30011)])
def test_values_only_requests_of_one_row_tile_per_wave_at_size(rt, npts, nreq):
    """Values of P5 / P4 tetrahedra at 74 / 75 points -- the one-row-tile instances of the request-per-workgroup kernel, 16-byte flush
    pieces with two workgroups per CU (56 rows x 74 points) and the 8-byte twin (35 x 75) -- over > 100 groups per workgroup, an odd
    number of requests: partition of unity over the whole batch, first / last / sampled requests against the C oracle."""
    import torch
    import fiat_amd
    from oracle import c_oracle
    el = fiat_amd.Lagrange(fiat_amd.ufc_simplex(3), 5 if npts == 74 else 4)
    ps = el.device_polyset()
    assert ps.kernel_name(0, nreq, npts, instance=True) == f"fxk::tabulate_simplex_wg<3,{5 if npts == 74 else 4},5>"
    rng = np.random.default_rng(npts)
    pts = simplex_points(rng, 3, (nreq, npts))
    out = ps.tabulate_batch(0, pts)
    assert float((out.sum(dim=2) - 1.0).abs().max()) <= 1e-11
    idx = np.concatenate([[0, 1, nreq - 2, nreq - 1], rng.choice(nreq, 200, replace=False)])
    got = out[torch.as_tensor(idx).cuda()].cpu().numpy()
    n = el.get_nodal_basis().get_embedded_degree()
    ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], n, el.get_coeffs(), 0, pts[idx], scale=el._expansion_scale,
                                  variant=el._expansion_variant).reshape(got.shape)
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
