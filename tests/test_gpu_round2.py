"""Round-2 device entries: fx_jacobi_batch, the kernel-selection policy, per-call scratch of the
shared-point path on concurrent streams, the work-queue check, and the RCCL gather behind the C ABI (world size 1
here: a one-GPU box; the multi-rank logic is covered on CPU by tests/test_distributed_cpu.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ab", [(0, 0), (1, 1), (2, 2), (3, 0), (5, 1)])
def test_jacobi_kernel_vs_reference(golden, ab):
    """Device tables equal FIAT's eval_jacobi_batch / eval_jacobi_deriv_batch (tests/golden/jacobi.npz, generated from
    the reference): 1e-12 relative on values, 1e-10 on derivatives."""
    from fiat_amd import jacobi
    g = golden("jacobi")
    a, b = ab
    xs = g["jacobi_x"]
    vals = jacobi.eval_jacobi_batch(a, b, 7, xs)
    ref = g[f"jacobi_{a}_{b}"]
    assert vals.shape == ref.shape
    assert np.abs(vals - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    der = jacobi.eval_jacobi_deriv_batch(a, b, 7, xs)
    dref = g[f"jacobi_deriv_{a}_{b}"]
    assert der.shape == dref.shape
    assert np.abs(der - dref).max() <= 1e-10 * max(1.0, np.abs(dref).max())
    # scalar twins (jacobi.py:15-44, :77-82)
    assert abs(jacobi.eval_jacobi(a, b, 5, 0.3) - float(jacobi.eval_jacobi_batch(a, b, 5, np.array([[0.3]]))[5, 0])) < 1e-14
    assert abs(jacobi.eval_jacobi_deriv(a, b, 5, 0.3) - float(jacobi.eval_jacobi_deriv_batch(a, b, 5, np.array([[0.3]]))[5, 0])) < 1e-13
    assert jacobi.eval_jacobi_deriv(a, b, 0, 0.3) == 0.0


def test_jacobi_kernel_higher_orders_and_large_batches(golden):
    """Derivative orders 2-4 (golden: the reference's own eval_jacobi_deriv_batch(order=...)), orders beyond the degree
    (all zero, jacobi.py:92-93), a batch of 1e6 points against the oracle's recurrence, degree 0."""
    from fiat_amd import jacobi
    from oracle import fiat_oracle as fo
    g = golden("round2")
    xs = g["jacobi_x"]
    for a, b in [(0, 0), (2, 1), (0.5, 1.5)]:
        for order in (2, 3, 4):
            ref = g[f"jacobi_deriv{order}_{a}_{b}"]
            got = jacobi.eval_jacobi_deriv_batch(a, b, 9, xs, order=order)
            assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (a, b, order)
    assert not jacobi.eval_jacobi_deriv_batch(1, 1, 3, xs, order=4).any()
    assert not jacobi.eval_jacobi_deriv_batch(1, 1, 3, xs, order=7).any()
    assert np.array_equal(jacobi.eval_jacobi_batch(2, 3, 0, xs), np.ones((1, len(xs))))
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=1_000_003)
    dev = jacobi.jacobi_table(1, 2, 12, torch.as_tensor(x).cuda())
    ref = fo.jacobi_table(1, 2, 12, x)
    assert np.abs(dev.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    with pytest.raises(ValueError):
        jacobi.jacobi_table(0, 0, -1, xs)
    with pytest.raises(NotImplementedError):
        jacobi.jacobi_table(0, 0, 200, xs)


def test_policy_reaches_every_kernel_family(golden, kernel_policy):
    """fx_ctx_set_policy replaces the per-launch environment switches: the same element and request shape run on the
    paired, K-streamed, LDS-image, stacked, cooperative and generic kernels and give the same tables."""
    from fiat_amd import runtime
    g = golden("elements")
    ps = runtime.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=g["c2_p3tet_q6_coeffs"])
    rng = np.random.default_rng(5)
    e = rng.exponential(size=(257, 23, 4))
    pts = (e / e.sum(-1, keepdims=True))[..., 1:].copy()
    ctx = runtime.Context.get()
    assert ctx.get_policy() == set()
    base = ps.tabulate_batch(1, pts).cpu().numpy()
    seen = {ps.kernel_name(1, 257, 23)}
    for names in (["kernel_stream"], ["kernel_image"], ["stacked_small"], ["no_fixed"], ["no_fixed", "no_stacked"],
                  ["no_fixed", "no_stacked", "no_coop", "no_small"]):
        kernel_policy(*names)
        assert ctx.get_policy() == set(names)
        seen.add(ps.kernel_name(1, 257, 23))
        out = ps.tabulate_batch(1, pts).cpu().numpy()
        assert np.abs(out - base).max() <= 1e-11 * max(1.0, np.abs(base).max()), names
    kernel_policy()
    assert {"fxk::tabulate_simplex_pair", "fxk::tabulate_simplex_stream", "fxk::tabulate_simplex_fixed",
            "fxk::tabulate_simplex_stacked", "fxk::tabulate_simplex_kernel"} <= seen, seen
    with pytest.raises(ValueError):
        kernel_policy("kernel_stream", "kernel_image")
    with pytest.raises(KeyError):
        kernel_policy("no_such_policy")
    with ctx.policy("no_stacked"):
        assert ctx.get_policy() == {"no_stacked"}
    assert ctx.get_policy() == set()


def test_shared_point_tabulation_on_concurrent_streams():
    """Two elements tabulated at one rule in many cells on two streams at once, plus a quadrature call in between: every
    call owns its reference-table scratch (stream-ordered allocation), so the results equal the serial ones."""
    import fiat_amd
    from fiat_amd import runtime
    rng = np.random.default_rng(9)
    sd, nreq, npts = 3, 20_000, 23
    ref = np.array(fiat_amd.ufc_simplex(sd).get_vertices(), dtype=float)
    verts = torch.as_tensor(ref[None] + rng.uniform(-0.15, 0.15, size=(nreq, sd + 1, sd))).cuda()
    e = rng.exponential(size=(2, npts, sd + 1))
    rules = [torch.as_tensor((x / x.sum(-1, keepdims=True))[:, 1:].copy()).cuda() for x in e]
    els = [fiat_amd.Lagrange(fiat_amd.ufc_simplex(sd), 3), fiat_amd.Nedelec(fiat_amd.ufc_simplex(sd), 2)]
    serial = [el.tabulate_cells(1, r, verts).clone() for el, r in zip(els, rules)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty_like(s) for s in serial]
    for rep in range(6):
        for el, r, s, o in zip(els, rules, streams, outs):
            with torch.cuda.stream(s):
                el.tabulate_cells(1, r, verts, out=o, stream=s)
        runtime.collapsed_quadrature(3, 4)
    torch.cuda.synchronize()
    for got, want in zip(outs, serial):
        assert torch.equal(got, want)
    runtime.Context.get().check()


def test_work_queue_check_is_clean_after_dynamic_launches(golden):
    from fiat_amd import runtime
    g = golden("elements")
    ps = runtime.SimplexPolySet(3, 3, variant="bubble", scale=1, coeffs=g["c2_p3tet_q6_coeffs"])
    rng = np.random.default_rng(1)
    e = rng.exponential(size=(30_001, 23, 4))
    out = ps.tabulate_batch(1, (e / e.sum(-1, keepdims=True))[..., 1:].copy())
    assert ps.kernel_name(1, 30_001, 23) == "fxk::tabulate_simplex_pair"
    host = runtime.fetch(out)          # .cpu() + fx_ctx_check
    assert np.isfinite(host).all()


@pytest.mark.parametrize("algo", ["direct", "ring"])
def test_rccl_gather_through_the_c_abi_world1(algo, tmp_path):
    """fx_comm_* / fx_allgather_tables with one rank: RCCL is loaded at run time, the communicator is created from the
    128-byte id, the block lands at recv[rank * stride + offset] (in place and out of place, chunked offsets)."""
    import torch.distributed as dist
    from fiat_amd import distributed as D
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)
    try:
        gather = D.TableGather(impl="rccl", algo=algo)
        assert gather.impl == "rccl" and gather.world == 1
        local = torch.arange(7 * 3 * 5, dtype=torch.float64, device="cuda").reshape(7, 3, 5)
        full = gather.all_gather(local, 7)
        torch.cuda.synchronize()
        assert torch.equal(full, local)
        # chunked, in place: rows produced chunk by chunk into the rank's block of the full buffer
        big = torch.full((10, 3, 5), float("nan"), dtype=torch.float64, device="cuda")

        def produce(lo, hi, rows):
            rows.copy_(local[lo:hi] if hi <= 7 else torch.cat([local[lo:7], torch.zeros(hi - 7, 3, 5, device="cuda", dtype=torch.float64)]))

        gather.tabulate_allgather(produce, 7, 10, 4, big)
        torch.cuda.synchronize()
        assert torch.equal(big[:7], local)
        seen = []
        for c0, c1, staged in gather.iter_gathered_chunks(local, 3, ring=2):
            seen.append(staged[0].clone())
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(seen), local)
        # the receive extent is validated behind the C ABI: a block that would end past the caller's buffer is refused
        import ctypes
        from fiat_amd import _lib, runtime
        small = torch.empty(5 * 3 * 5, dtype=torch.float64, device="cuda")
        rc = _lib.lib.fx_allgather_tables(gather.comm, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(small.data_ptr()),
                                          local.numel(), local.numel(), 0, small.numel(), 1, runtime._stream_ptr(None))
        assert rc != 0 and b"exceed the receive buffer" in _lib.lib.fx_last_error()
        with pytest.raises(ValueError):
            gather._exchange(local, small.view(5, 3, 5), 7, 0)
        # close() drains the stream the last exchange went to before the communicator is destroyed
        out = gather.all_gather(local, 7)
        gather.close()
        assert torch.equal(out, local) and gather.comm is None
    finally:
        dist.destroy_process_group()


# ---- derivative orders 3 and 4 (differentiation-matrix route, csrc/api.hip ensure_high_order) ---------------------
def _rel(x, ref):
    return np.abs(x - ref).max() / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("sd", [1, 2, 3])
@pytest.mark.parametrize("variant", [None, "bubble"])
@pytest.mark.parametrize("n", [2, 4])
def test_expansion_sets_to_order_4(golden, sd, variant, n):
    """ExpansionSet._tabulate(n, pts, order=4) against the reference (FIAT/expansions.py:66-137 at orders 3 and 4; the
    relation test/FIAT/unit/test_polynomial.py:87-109 checks to 1e-10); orders above the degree are exact zeros there."""
    import fiat_amd
    g = golden("round2")
    es = fiat_amd.ExpansionSet(fiat_amd.ufc_simplex(sd), variant=variant)
    tab = es._tabulate(n, g[f"hi_pts_sd{sd}"], order=4)
    ref = g[f"hi_exp_sd{sd}_{variant}_n{n}"]
    keys = [a for k in range(5) for a in fiat_amd.mis(sd, k)]
    assert list(tab) == keys
    for t, a in enumerate(keys):
        tol = 1e-12 if sum(a) == 0 else 1e-10
        assert _rel(tab[a], ref[t]) <= tol, (a, _rel(tab[a], ref[t]))


@pytest.mark.parametrize("name,make,sd", [("p3tet", lambda fa, c: fa.Lagrange(c, 3), 3), ("dg4tri", lambda fa, c: fa.DiscontinuousLagrange(c, 4), 2),
                                          ("n2tet", lambda fa, c: fa.Nedelec(c, 2), 3), ("p5line", lambda fa, c: fa.Lagrange(c, 5), 1),
                                          ("rt3tri", lambda fa, c: fa.RaviartThomas(c, 3), 2),
                                          ("dg6tet", lambda fa, c: fa.DiscontinuousLagrange(c, 6), 3)])
def test_elements_to_order_4(golden, name, make, sd):
    """element.tabulate(3, .) and (4, .) -- what test/FIAT/regression/test_regression.py:283-299 tabulates -- against
    the reference, single call and batched (two requests, ragged to the kernels: 9 points)."""
    import fiat_amd
    g = golden("round2")
    el = make(fiat_amd, fiat_amd.ufc_simplex(sd))
    pts = g[f"hi_pts_sd{sd}"]
    for order in (3, 4):
        ref = g[f"hi_{name}_o{order}"]
        tab = el.tabulate(order, pts)
        keys = [a for k in range(order + 1) for a in fiat_amd.mis(sd, k)]
        assert list(tab) == keys
        for t, a in enumerate(keys):
            assert tab[a].shape == ref[t].shape
            assert _rel(tab[a], ref[t]) <= (1e-12 if t == 0 else 1e-10), (order, a, _rel(tab[a], ref[t]))
        if sd > 1:
            dev = el.tabulate_batch(order, np.stack([pts, pts[::-1]])).cpu().numpy()
            assert _rel(dev[0], ref) <= 1e-10 and _rel(dev[1][..., ::-1], ref) <= 1e-10


def test_order_3_on_a_physical_cell_and_limits(golden):
    import fiat_amd
    g = golden("round2")
    cell = fiat_amd.physical_simplex(g["hi_phys_verts"])
    el = fiat_amd.Lagrange(cell, 3)
    tab = el.tabulate(3, g["hi_phys_pts"])
    ref = g["hi_phys_p3tet_o3"]
    for t, a in enumerate([a for k in range(4) for a in fiat_amd.mis(3, k)]):
        assert _rel(tab[a], ref[t]) <= (1e-12 if t == 0 else 1e-10), (a, _rel(tab[a], ref[t]))
    # all third derivatives of a cubic are constant over the points, fourth derivatives vanish
    t4 = el.tabulate(4, g["hi_phys_pts"])
    for a in fiat_amd.mis(3, 3):
        assert np.abs(t4[a] - t4[a][:, :1]).max() <= 1e-9 * max(1.0, np.abs(t4[a]).max())
    for a in fiat_amd.mis(3, 4):
        assert np.abs(t4[a]).max() <= 1e-8
    with pytest.raises(NotImplementedError):
        el.tabulate(9, g["hi_phys_pts"])
    # per-request cells: any order the own-cell route serves (round 4: table_mix_any_kernel; tests/test_gpu_round4.py); all
    # derivatives of a cubic beyond the third vanish on the request's cell too
    t5 = el.tabulate_batch(5, g["hi_phys_pts"][None], verts=g["hi_phys_verts"][None]).cpu().numpy()[0]
    n3 = sum(len(fiat_amd.mis(3, k)) for k in range(4))
    assert np.abs(t5[n3:]).max() <= 1e-6 and np.abs(t5[:n3] - np.stack([tab[a] for k in range(4) for a in fiat_amd.mis(3, k)])).max() <= 1e-8
    with pytest.raises(NotImplementedError):
        el.tabulate_batch(9, g["hi_phys_pts"][None], verts=g["hi_phys_verts"][None])


def _chain_rule_tables(fa, ref_tab, sd, order, Kt):
    """Derivatives with respect to x from the tables with respect to X (reference cell), Kt[c, d] = dX_c / dx_d:
    d^alpha_x = sum over ordered source directions of prod Kt[c_m, d_m] d^beta_X -- NumPy, independent of the device pass."""
    import itertools
    out = []
    for alpha in [a for k in range(order + 1) for a in fa.mis(sd, k)]:
        dirs = [d for d, m in enumerate(alpha) for _ in range(m)]
        acc = 0.0
        for src in itertools.product(range(sd), repeat=len(dirs)):
            beta = tuple(src.count(c) for c in range(sd))
            acc = acc + float(np.prod([Kt[c, d] for c, d in zip(src, dirs)])) * ref_tab[beta]
        out.append(acc)
    return np.stack(out)


@pytest.mark.parametrize("order", [3, 4])
def test_high_orders_with_per_request_cells(golden, order):
    """Orders 3 and 4 with per-request cells (table_mix_high_kernel: symmetric tensor powers of K = dX/dx across the tables).
    (a) The reference's P3 element built ON a physical tetrahedron (round2.npz) equals the UFC-cell element tabulated with that
    cell as the request's cell; (b) random cells incl. a negatively oriented one, sd = 1, 2, 3, scalar and vector-valued sets:
    equal to the chain rule applied in NumPy to the reference-cell tables, and -- affine families -- to elements built
    directly on the physical cells."""
    import fiat_amd as fa
    g = golden("round2")
    if order == 3:
        el = fa.Lagrange(fa.ufc_simplex(3), 3)
        got = el.tabulate_batch(3, g["hi_phys_pts"][None], verts=g["hi_phys_verts"][None]).cpu().numpy()[0]
        for t in range(got.shape[0]):
            assert _rel(got[t], g["hi_phys_p3tet_o3"][t]) <= (1e-12 if t == 0 else 1e-10), t
    rng = np.random.default_rng(40 + order)
    cases = [(3, lambda c: fa.Lagrange(c, 4), True), (3, lambda c: fa.DiscontinuousLagrange(c, 5), True),
             (2, lambda c: fa.Lagrange(c, 5), True), (2, lambda c: fa.RaviartThomas(c, 3), False),
             (1, lambda c: fa.ONPolynomialSet(c, 6), False)]
    for sd, make, rebuild in cases:
        ref = np.array(fa.ufc_simplex(sd).get_vertices(), dtype=float)
        ncell, npts = 3, 11
        A = np.eye(sd) + 0.25 * rng.standard_normal((ncell, sd, sd))
        A[-1, :, 0] *= -1.0                                          # one negatively oriented cell
        verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((ncell, 1, sd))
        e = rng.exponential(size=(ncell, npts, sd + 1))
        bary = e / e.sum(-1, keepdims=True)
        pts, ref_pts = np.einsum("rpv,rvd->rpd", bary, verts), np.einsum("rpv,vd->rpd", bary, ref)
        base = make(fa.ufc_simplex(sd))
        is_element = hasattr(base, "dual_basis")
        dev = base if is_element else base.device_polyset()
        got = dev.tabulate_batch(order, pts, verts=verts).cpu().numpy()
        for r in range(ncell):
            ref_tab = base.tabulate(order, ref_pts[r]) if is_element else base.tabulate(ref_pts[r], order)
            J = (verts[r][1:] - verts[r][0]).T @ np.linalg.inv((ref[1:] - ref[0]).T)          # dx/dX
            want = _chain_rule_tables(fa, ref_tab, sd, order, np.linalg.inv(J))
            assert got[r].shape == want.shape
            assert _rel(got[r], want) <= 1e-9, (sd, r, _rel(got[r], want))
            if rebuild:
                tab = make(fa.physical_simplex(verts[r])).tabulate(order, pts[r])
                direct = np.stack([tab[a] for k in range(order + 1) for a in fa.mis(sd, k)])
                assert _rel(got[r], direct) <= 1e-9, (sd, r)


# ---- general tensor products (csrc/table_kernels.hpp table_outer_kernel) ----------------------------------------
def _tp(fa, key):
    T, I = fa.ufc_simplex(2), fa.ufc_simplex(1)
    return {"p2tri_p1": lambda: fa.TensorProductElement(fa.Lagrange(T, 2), fa.Lagrange(I, 1)),
            "dg1tri_p2": lambda: fa.TensorProductElement(fa.DiscontinuousLagrange(T, 1), fa.Lagrange(I, 2)),
            "rt1tri_dg0": lambda: fa.TensorProductElement(fa.RaviartThomas(T, 1), fa.DiscontinuousLagrange(I, 0)),
            "n1tri_p1": lambda: fa.TensorProductElement(fa.Nedelec(T, 1), fa.Lagrange(I, 1)),
            "rt2tri_dg1": lambda: fa.TensorProductElement(fa.RaviartThomas(T, 2), fa.DiscontinuousLagrange(I, 1))}[key]()


@pytest.mark.parametrize("key", ["p2tri_p1", "dg1tri_p2", "rt1tri_dg0", "n1tri_p1", "rt2tri_dg1"])
def test_prism_elements_vs_reference(golden, key):
    """Triangle x interval products, scalar and with a vector-valued triangle factor (FIAT/tensor_product.py:274-317),
    orders 0-2, against the reference; batched = single."""
    import fiat_amd
    g = golden("round2")
    el = _tp(fiat_amd, key)
    pts = g["tp_prism_pts"]
    for order in (0, 1, 2):
        ref = g[f"tp_{key}_o{order}"]
        tab = el.tabulate(order, pts)
        keys = [a for k in range(order + 1) for a in fiat_amd.mis(3, k)]
        assert list(tab) == keys
        for t, a in enumerate(keys):
            assert tab[a].shape == ref[t].shape, (tab[a].shape, ref[t].shape)
            assert _rel(tab[a], ref[t]) <= (1e-12 if t == 0 else 1e-10), (order, a)
    dev = el.tabulate_batch(1, np.stack([pts] * 3)).cpu().numpy()
    assert _rel(dev[2], g[f"tp_{key}_o1"]) <= 1e-10
    assert el.space_dimension() == ref.shape[1]
    assert el.value_shape() == (() if ref.ndim == 3 else (2,))


def test_scalar_times_vector_and_entities(golden):
    import fiat_amd as fa
    g = golden("round2")
    T, I = fa.ufc_simplex(2), fa.ufc_simplex(1)
    for key, el in (("tp_p1_rt1tri_o1", fa.TensorProductElement(fa.Lagrange(I, 1), fa.RaviartThomas(T, 1))),
                    ("tp_p2_dg1tri_o1", fa.TensorProductElement(fa.Lagrange(I, 2), fa.DiscontinuousLagrange(T, 1)))):
        tab = el.tabulate(1, g["tp_it_pts"])
        for t, a in enumerate([a for k in range(2) for a in fa.mis(3, k)]):
            assert tab[a].shape == g[key][t].shape
            assert _rel(tab[a], g[key][t]) <= 1e-10, (key, a)
    el = fa.TensorProductElement(fa.Lagrange(T, 2), fa.Lagrange(I, 1))
    keys = [a for k in range(2) for a in fa.mis(3, k)]
    for k in (0, 1):       # bottom / top triangle
        tab = el.tabulate(1, g["tp_tri_pts"], entity=((2, 0), k))
        for t, a in enumerate(keys):
            assert _rel(tab[a], g[f"tp_p2tri_p1_ent20_{k}"][t]) <= 1e-10, (k, a)
    for k in (0, 1, 2):    # side quadrilaterals
        tab = el.tabulate(1, g["tp_quad_pts"], entity=((1, 1), k))
        for t, a in enumerate(keys):
            assert _rel(tab[a], g[f"tp_p2tri_p1_ent11_{k}"][t]) <= 1e-10, (k, a)
    with pytest.raises(NotImplementedError):
        fa.TensorProductElement(fa.RaviartThomas(T, 1), fa.RaviartThomas(T, 1))
    # the hexahedron of BASELINE config 5 stays on the fused kernel; the general route gives the same tables
    P2 = fa.Lagrange(I, 2)
    hexel = fa.TensorProductElement(fa.TensorProductElement(P2, P2), P2)
    pts = np.random.default_rng(3).uniform(0, 1, size=(2, 11, 3))
    fused = hexel.tabulate_batch(1, pts)
    left = hexel.A.tabulate_batch(1, pts[..., :2])
    right = P2.tabulate_batch(1, pts[..., 2:])
    from fiat_amd import runtime
    assert _rel(runtime.table_outer(1, 2, 1, left, right).cpu().numpy(), fused.cpu().numpy()) <= 1e-13


# ---- sub-entity tabulation on the device (fx_map_points) ------------------------------------------------------------
def test_entity_tabulation_vs_reference(golden):
    """tabulate(order, points, entity=(dim, id)): facet, edge and vertex entities of a tetrahedron, edges of a triangle
    for a Piola-mapped element (FIAT/finite_element.py:181-197, reference_element.py:570-609); batched facet tabulation
    and one facet rule in many cells."""
    import fiat_amd as fa
    g = golden("round2")
    el = fa.Lagrange(fa.ufc_simplex(3), 3)
    keys = [a for k in range(2) for a in fa.mis(3, k)]
    for f in range(4):
        tab = el.tabulate(1, g["ent_facet_pts"], entity=(2, f))
        for t, a in enumerate(keys):
            assert _rel(tab[a], g[f"ent_p3tet_facet{f}"][t]) <= 1e-10, (f, a)
    for e in range(6):
        tab = el.tabulate(1, g["ent_edge_pts"], entity=(1, e))
        for t, a in enumerate(keys):
            assert _rel(tab[a], g[f"ent_p3tet_edge{e}"][t]) <= 1e-10, (e, a)
    tab = el.tabulate(1, np.zeros((1, 0)), entity=(0, 2))
    for t, a in enumerate(keys):
        assert tab[a].shape == g["ent_p3tet_vertex2"][t].shape
        assert _rel(tab[a], g["ent_p3tet_vertex2"][t]) <= 1e-10
    rt = fa.RaviartThomas(fa.ufc_simplex(2), 2)
    for e in range(3):
        tab = rt.tabulate(1, g["ent_edge_pts"], entity=(1, e))
        for t, a in enumerate([a for k in range(2) for a in fa.mis(2, k)]):
            assert _rel(tab[a], g[f"ent_rt2tri_edge{e}"][t]) <= 1e-10, (e, a)
    # batch of facet point sets, and a facet quadrature rule pushed to many physical cells
    batch = np.stack([g["ent_facet_pts"], g["ent_facet_pts"][::-1]])
    dev = el.tabulate_batch(1, batch, entity=(2, 3)).cpu().numpy()
    assert _rel(dev[0], g["ent_p3tet_facet3"]) <= 1e-10 and _rel(dev[1][..., ::-1], g["ent_p3tet_facet3"]) <= 1e-10
    Q = fa.create_quadrature(fa.ufc_simplex(2), 4)
    rng = np.random.default_rng(4)
    ref = np.array(fa.ufc_simplex(3).get_vertices(), dtype=float)
    verts = ref[None] + rng.uniform(-0.1, 0.1, size=(50, 4, 3))
    got = el.tabulate_cells(1, Q.device_points()[0], verts, entity=(2, 1)).cpu().numpy()
    M, b = el.entity_map((2, 1))
    ref_pts = Q.get_points() @ M.T + b
    bary = np.concatenate([1 - ref_pts.sum(1, keepdims=True), ref_pts], axis=1)
    phys = np.einsum("pv,rvd->rpd", bary, verts)
    want = el.tabulate_batch(1, phys, verts=verts).cpu().numpy()
    assert _rel(got, want) <= 1e-10


def test_default_rule_on_device_feeds_tabulate_cells(golden):
    """create_quadrature(tet, 6) is the 23-point Xiao-Gimbutas rule of BASELINE config 2; resident on the device it feeds
    tabulate_cells, whose tables on the reference cell equal the reference's c2_p3tet_q6_tab."""
    import fiat_amd as fa
    g = golden("elements")
    Q = fa.create_quadrature(fa.ufc_simplex(3), 6)
    pts_d, wts_d = Q.device_points()
    assert pts_d.is_cuda and pts_d.shape == (23, 3) and abs(float(wts_d.sum()) - 1 / 6) < 1e-14
    el = fa.Lagrange(fa.ufc_simplex(3), 3)
    ref = np.array(fa.ufc_simplex(3).get_vertices(), dtype=float)
    out = el.tabulate_cells(1, pts_d, np.stack([ref, ref])).cpu().numpy()
    assert _rel(out[0], g["c2_p3tet_q6_tab"]) <= 1e-10 and _rel(out[1], g["c2_p3tet_q6_tab"]) <= 1e-10


def test_third_order_derivative_functionals():
    """DualSet.to_riesz with derivative functionals of order 3 (FIAT/dual_set.py:175-205 at orders the round-1 device path
    refused): the septic Hermite element on the interval -- value and the first three derivatives at both end points -- built
    through the device Riesz assembly.  Independent check: the basis function dual to the value at 0 is
    (1 - x)^4 (1 + 4x + 10x^2 + 20x^3); every dof applied to every basis function is the identity."""
    import fiat_amd as fa
    from fiat_amd import dual_set, finite_element, functional, polynomial_set
    cell = fa.ufc_simplex(1)
    nodes = []
    for x in (0.0, 1.0):
        nodes.append(functional.PointEvaluation(cell, (x,)))
        nodes += [functional.PointDerivative(cell, (x,), (k,)) for k in (1, 2, 3)]
    ids = {0: {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}, 1: {0: []}}
    el = finite_element.CiarletElement(polynomial_set.ONPolynomialSet(cell, 7), dual_set.DualSet(nodes, cell, ids), 7)
    x = np.linspace(0.05, 0.95, 13)
    tab = el.tabulate(3, x[:, None])
    want = (1 - x) ** 4 * (1 + 4 * x + 10 * x ** 2 + 20 * x ** 3)
    assert np.abs(tab[(0,)][0] - want).max() <= 1e-12
    ends = el.tabulate(3, np.array([[0.0], [1.0]]))
    V = np.stack([ends[(k,)][:, e] for e in (0, 1) for k in (0, 1, 2, 3)])       # dof i applied to basis function j
    assert np.abs(V - np.eye(8)).max() <= 1e-9


# ---- a list of heterogeneous requests (fiat_amd/batch.py) ---------------------------------------------------------------
def test_request_list_is_grouped_and_batched():
    """600 requests over seven elements, orders 0-2, 3-23 points, some on their own cells, in random order: every result equals
    the one-request-at-a-time ``element.tabulate`` / ``tabulate_batch`` of the same request; results are views of a few group
    tensors; with rank / world each rank returns its contiguous share of every group."""
    import fiat_amd as fa
    rng = np.random.default_rng(99)
    T2, T3 = fa.ufc_simplex(2), fa.ufc_simplex(3)
    elements = [fa.Lagrange(T2, 1), fa.Lagrange(T2, 3), fa.Lagrange(T3, 2), fa.Lagrange(T3, 3), fa.DiscontinuousLagrange(T3, 1),
                fa.Nedelec(T3, 1), fa.RaviartThomas(T2, 2)]
    requests = []
    for _ in range(600):
        el = elements[rng.integers(len(elements))]
        sd = el.get_reference_element().get_spatial_dimension()
        npts = int(rng.choice([3, 6, 11, 23]))
        e = rng.exponential(size=(npts, sd + 1))
        bary = e / e.sum(-1, keepdims=True)
        ref = np.array(el.get_reference_element().get_vertices(), dtype=float)
        if rng.random() < 0.3:
            verts = ref @ (np.eye(sd) + 0.2 * rng.standard_normal((sd, sd))).T + rng.standard_normal(sd)
            requests.append(fa.Request(el, int(rng.integers(0, 3)), bary @ verts, verts))
        else:
            requests.append(fa.Request(el, int(rng.integers(0, 3)), bary @ ref))
    results = fa.tabulate_requests(requests)
    torch.cuda.synchronize()
    assert len(results) == 600 and all(r is not None for r in results)
    assert len({r.untyped_storage().data_ptr() for r in results}) < 200       # views of the group tensors, not 600 allocations
    for req, got in zip(requests[::7], results[::7]):
        sd = req.element.get_reference_element().get_spatial_dimension()
        if req.verts is None:
            tab = req.element.tabulate(req.order, req.points)
            want = np.stack([tab[a] for k in range(req.order + 1) for a in fa.mis(sd, k)])
        else:
            want = req.element.tabulate_batch(req.order, req.points[None], verts=req.verts[None]).cpu().numpy()[0]
        assert got.shape == want.shape
        assert _rel(got.cpu().numpy(), want) <= 1e-10
    # two ranks: disjoint shares covering everything, equal to the single-rank results
    halves = [fa.tabulate_requests(requests, rank=r, world=2) for r in (0, 1)]
    torch.cuda.synchronize()
    for i in range(600):
        owners = [h[i] for h in halves if h[i] is not None]
        assert len(owners) == 1 and torch.equal(owners[0], results[i])


def test_hexahedron_facets_and_edges(golden):
    """Nested products: the entity dimensions of (P2 x P1) x P2 are nested tuples; faces ((1,1),0), ((1,0),1), ((0,1),1) and
    edges ((1,0),0) of the hexahedron against the reference (FIAT/tensor_product.py:231-258 unravels the entity number over the
    factors' entities)."""
    import fiat_amd as fa
    g = golden("round2")
    I = fa.ufc_simplex(1)
    P2 = fa.Lagrange(I, 2)
    hexel = fa.TensorProductElement(fa.TensorProductElement(P2, fa.Lagrange(I, 1)), P2)
    assert hexel.get_reference_element().get_dimension() == ((1, 1), 1)
    keys = [a for k in range(2) for a in fa.mis(3, k)]
    quad = g["tp_hex_quad_pts"]
    for dims, count in ((((1, 1), 0), 2), (((1, 0), 1), 2), (((0, 1), 1), 2), (((1, 0), 0), 4)):
        pts = quad if sum(dims[0]) + dims[1] == 2 else quad[:, :1]
        name = "".join(str(x) for x in (*dims[0], dims[1]))
        for k in range(count):
            tab = hexel.tabulate(1, pts, entity=(dims, k))
            ref = g[f"tp_hex_ent{name}_{k}"]
            for t, a in enumerate(keys):
                assert tab[a].shape == ref[t].shape
                assert _rel(tab[a], ref[t]) <= 1e-10, (dims, k, a)
    # the cell itself still runs on the fused kernel and agrees with the general route
    pts = np.random.default_rng(8).uniform(0, 1, size=(9, 3))
    whole = hexel.tabulate(1, pts)
    same = hexel.tabulate(1, pts, entity=(((1, 1), 1), 0))
    for a in keys:
        assert _rel(whole[a], same[a]) <= 1e-13


def test_jacobi_kernel_extremes():
    """Chebyshev weights (a = b = -1/2), the highest degree the kernel-argument table takes (96) with derivative order 8, against
    the oracle's recurrence / the closed form T_n(cos t) = cos(n t)."""
    from fiat_amd import jacobi
    from oracle import fiat_oracle as fo
    t = np.linspace(0.05, 3.0, 41)
    x = np.cos(t)
    tab = jacobi.eval_jacobi_batch(-0.5, -0.5, 20, x[:, None])
    for n in (1, 5, 20):      # P_n^(-1/2,-1/2) = binom(n - 1/2, n) T_n
        scale = np.prod([(k - 0.5) / k for k in range(1, n + 1)])
        assert np.abs(tab[n] - scale * np.cos(n * t)).max() <= 1e-12
    xs = np.linspace(-1, 1, 17)
    dev = jacobi.eval_jacobi_deriv_batch(1.0, 2.0, 96, xs[:, None], order=8)
    ref = fo.jacobi_deriv_table(1.0, 2.0, 96, xs, order=8)
    assert dev.shape == ref.shape == (97, 17)
    assert np.abs(dev - ref).max() <= 1e-10 * np.abs(ref).max()


def test_flattened_hexahedron_entities(golden):
    """FlattenedDimensions: faces (2, 0..5) and edges (1, 0..11) of the hexahedron map to the product entities as in the
    reference (FIAT/tensor_product.py:396-407, reference_element.py:1852-1866)."""
    import fiat_amd as fa
    g = golden("round2")
    I = fa.ufc_simplex(1)
    P2 = fa.Lagrange(I, 2)
    flat = fa.FlattenedDimensions(fa.TensorProductElement(fa.TensorProductElement(P2, fa.Lagrange(I, 1)), P2))
    keys = [a for k in range(2) for a in fa.mis(3, k)]
    quad = g["tp_hex_quad_pts"]
    for k in range(6):
        tab = flat.tabulate(1, quad, entity=(2, k))
        for t, a in enumerate(keys):
            assert _rel(tab[a], g[f"tp_flathex_face{k}"][t]) <= 1e-10, (k, a)
    for k in range(12):
        tab = flat.tabulate(1, quad[:, :1], entity=(1, k))
        for t, a in enumerate(keys):
            assert _rel(tab[a], g[f"tp_flathex_edge{k}"][t]) <= 1e-10, (k, a)
    tab = flat.tabulate(1, g["tp_prism_pts"])
    for t, a in enumerate(keys):
        assert _rel(tab[a], g["tp_flathex_cell"][t]) <= 1e-10
    # order 3 on the hexahedron: beyond the fused kernel, through the factor tables (1-D Lagrange derivatives of any order)
    tab3 = flat.tabulate(3, g["tp_prism_pts"])
    for t, a in enumerate([a for k in range(4) for a in fa.mis(3, k)]):
        assert _rel(tab3[a], g["tp_hex_o3"][t]) <= 1e-10, a
    dofs = flat.entity_dofs()
    assert sorted(dofs) == [0, 1, 2, 3] and [len(dofs[d]) for d in range(4)] == [8, 12, 6, 1]
    assert sum(len(v) for d in dofs.values() for v in d.values()) == flat.space_dimension() == 18


def test_random_shapes_default_kernels_match_generic():
    """tools/fuzz_policies.py for a few seconds: random elements / orders / point counts / batch sizes / per-request cells,
    default kernel selection against the generic kernel (a 300 s run covered 19 501 simplex and 10 053 tensor / prism cases, worst difference 5e-12)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_policies.py"), "8", "5"], capture_output=True,
                         text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "MISMATCH" not in out.stdout and "random cases" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("n,npts", [(3, 17), (3, 23), (3, 32), (3, 33), (3, 48), (4, 23), (4, 24), (4, 31), (4, 33), (4, 44), (4, 47)])
@pytest.mark.parametrize("cells", [False, True])
def test_values_only_paired_instances_vs_c_oracle(n, npts, cells):
    """Values-only (order 0) instances of the paired kernel for P3 / P4 tetrahedra.  35 rows x an odd point count is an odd
    number of doubles per request: the 16-byte flush cannot serve it, and the selection must send it elsewhere (a first
    version of the P4 instances lacked that guard and wrote wrong tables for odd point counts)."""
    import math
    from oracle import c_oracle, fiat_oracle as fo
    from fiat_amd import runtime
    rng = np.random.default_rng(100 * n + npts)
    nexp = math.comb(n + 3, 3)
    co = rng.standard_normal((nexp, nexp))
    ps = runtime.SimplexPolySet(3, n, coeffs=co)
    ref_cell = fo.UFC_SIMPLEX[3]
    for nreq in (1, 2, 5, 257):
        e = rng.exponential(size=(nreq, npts, 4))
        bary = e / e.sum(-1, keepdims=True)
        verts = None
        pts = np.einsum("rpv,vd->rpd", bary, ref_cell)
        if cells:
            A = np.eye(3) + 0.2 * rng.standard_normal((nreq, 3, 3))
            verts = np.einsum("vd,red->rve", ref_cell, A) + rng.standard_normal((nreq, 1, 3))
            pts = np.einsum("rpv,rvd->rpd", bary, verts)
        kern = ps.kernel_name(0, nreq, npts, has_verts=cells)
        if (nexp * npts) % 2 == 0 and 17 <= npts <= 48:
            assert kern == "fxk::tabulate_simplex_pair", kern
        else:
            assert kern != "fxk::tabulate_simplex_pair", kern
        out = ps.tabulate_batch(0, pts, verts=verts).cpu().numpy()
        ref = c_oracle.tabulate_batch(ref_cell, n, co, 0, pts, verts=verts).reshape(out.shape)
        err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 1e-12, (n, npts, nreq, cells, kern, err)


def _tensor_oracle(nodes, dim, order, pts):
    from oracle import fiat_oracle as fo
    t = [fo.lagrange_line_tabulate(nodes, pts[:, d:d + 1], order) for d in range(dim)]
    tab = fo.tensor_product_tabulate(t[0], 1, t[1], 1, order)
    if dim == 3:
        tab = fo.tensor_product_tabulate(tab, 2, t[2], 1, order)
    return np.stack([tab[a] for a in fo.jet_indices(dim, order)])


@pytest.mark.parametrize("dim,nn,order", [(2, 2, 0), (2, 2, 1), (2, 2, 2), (2, 3, 0), (2, 3, 1), (2, 3, 2), (2, 4, 0), (2, 4, 1), (2, 4, 2),
                                          (2, 5, 0), (2, 5, 1), (3, 2, 0), (3, 2, 1), (3, 2, 2), (3, 3, 0)])
def test_lane_local_tensor_kernel_vs_oracle(dim, nn, order):
    """tensor_small_kernel (Q1..Q4 quadrilaterals, Q1 / Q2 hexahedra: requests up to 16 KB) against the restated
    TensorProductElement.tabulate (FIAT/tensor_product.py:231-292 over barycentric_interpolation.py:22-93): scattered points and
    tensor grids, ragged batches, a point on a node (exact Kronecker delta), and equality with the one-workgroup-per-request kernel."""
    from fiat_amd import runtime
    ctx = runtime.Context.get()
    rng = np.random.default_rng(1000 * dim + 10 * nn + order)
    nodes = np.concatenate([[0.0, 1.0], np.linspace(0, 1, nn)[1:-1]])          # FIAT's entity order: vertices first
    L = runtime.LineLagrange(nodes)
    for q in (1, 2, nn, nn + 1):
        npts = q ** dim
        if npts > 64:
            continue
        for nreq in (1, 3, 17, 1000):
            grid = rng.uniform(0, 1, size=(nreq, dim, q))
            grid[0, 0, 0] = nodes[-1]                                               # a coordinate exactly on a node
            pts = np.stack([np.stack(np.meshgrid(*g, indexing="ij"), axis=-1).reshape(-1, dim) for g in grid])
            a = runtime.tensor_tabulate_batch([L] * dim, order, grid, grid=True).cpu().numpy()
            b = runtime.tensor_tabulate_batch([L] * dim, order, pts).cpu().numpy()
            assert np.array_equal(a, b)
            ctx.set_policy("no_small")
            try:
                c = runtime.tensor_tabulate_batch([L] * dim, order, pts).cpu().numpy()
            finally:
                ctx.set_policy()
            assert np.array_equal(a, c), "lane-local and per-request kernels multiply the same factors"
            for r in sorted({0, nreq // 2, nreq - 1}):
                ref = _tensor_oracle(nodes, dim, order, pts[r])
                err = np.abs(a[r] - ref).max(axis=(1, 2)) / np.maximum(1.0, np.abs(ref).max(axis=(1, 2)))
                assert err[0] <= 1e-12 and err.max() <= 1e-10, (dim, nn, order, q, nreq, r, err)
    ctx.check()


@pytest.mark.parametrize("family,sd,deg,qdeg", [("Lagrange", 2, 1, 2), ("Lagrange", 2, 5, 10), ("RaviartThomas", 3, 2, 4),
                                                 ("Nedelec", 3, 3, 6)])
def test_one_rule_many_cells_with_hessians_odd_tables(family, sd, deg, qdeg):
    """tabulate_cells (fx_tabulate_batch_shared) at order 2 for tables with an odd number of doubles (rows x points): these ran on
    the one-workgroup-per-request fallback; now the register-resident kernel with one double per slot.  Reference: the
    per-request-point path with the same cells and the rule mapped into them (itself pinned against the reference elements
    built on physical cells, tests/test_gpu_pushforward.py)."""
    import fiat_amd
    cell = fiat_amd.ufc_simplex(sd)
    el = getattr(fiat_amd, family)(cell, deg)
    rule = np.asarray(fiat_amd.create_quadrature(cell, qdeg).get_points())
    rows = el.space_dimension() * int(np.prod(el.value_shape() or (1,)))
    assert (rows * len(rule)) % 2 == 1
    rng = np.random.default_rng(7)
    ref = np.array(cell.get_vertices(), dtype=float)
    for nreq in (1, 37, 1000):
        A = np.eye(sd) + 0.2 * rng.standard_normal((nreq, sd, sd))
        verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((nreq, 1, sd))
        # the rule's points in every physical cell: x = v0 + sum_i xi_i (v_i - v0) on the UFC simplex
        pts = verts[:, :1, :] + np.einsum("pi,rid->rpd", rule, verts[:, 1:, :] - verts[:, :1, :])
        for order in (0, 1, 2):
            a = el.tabulate_cells(order, rule, verts).cpu().numpy()
            b = el.tabulate_batch(order, pts, verts=verts, pushforward=True).cpu().numpy()
            axes = tuple(range(2, a.ndim))
            err = (np.abs(a - b).max(axis=axes) / np.maximum(1.0, np.abs(b).max(axis=axes))).max()
            assert err <= 1e-10, (family, sd, deg, nreq, order, err)


@pytest.mark.parametrize("fa,ka,fb,kb", [("Lagrange", 1, "Lagrange", 1), ("Lagrange", 2, "Lagrange", 1), ("Lagrange", 2, "Lagrange", 2),
                                         ("Lagrange", 3, "Lagrange", 2), ("DiscontinuousLagrange", 1, "DiscontinuousLagrange", 1),
                                         ("RaviartThomas", 1, "DiscontinuousLagrange", 0), ("Nedelec", 1, "Lagrange", 1),
                                         ("BrezziDouglasMarini", 1, "Lagrange", 3), ("Lagrange", 3, "Lagrange", 3)])
def test_fused_prism_kernel_matches_general_route(fa, ka, fb, kb):
    """prism_small_kernel (triangle factor x 1-D Lagrange factor in one pass) against the general route (two factor
    tabulations + fx_table_outer_batch, itself pinned by the reference's prism tables in round2.npz): orders 0-2, ragged
    batches, scalar and vector-valued triangle factors, the constant on the interval."""
    import fiat_amd
    from fiat_amd import runtime
    ctx = runtime.Context.get()
    A = getattr(fiat_amd, fa)(fiat_amd.ufc_simplex(2), ka)
    B = getattr(fiat_amd, fb)(fiat_amd.ufc_simplex(1), kb)
    el = fiat_amd.TensorProductElement(A, B)
    prism = el._prism_factors()
    assert prism is not None
    rng = np.random.default_rng(17)
    for order in (0, 1, 2):
        for nreq, npts in ((1, 1), (3, 6), (50, 18), (1000, 7)):
            e = rng.exponential(size=(nreq, npts, 3))
            xy = (e / e.sum(-1, keepdims=True))[..., 1:]
            pts = np.concatenate([xy, rng.uniform(0, 1, size=(nreq, npts, 1))], axis=-1)
            fused = runtime.prism_tabulate_batch(prism[0], prism[1], order, pts)
            a = el.tabulate_batch(order, pts).cpu().numpy()
            ctx.set_policy("no_small")
            try:
                assert runtime.prism_tabulate_batch(prism[0], prism[1], order, pts) is None
                b = el.tabulate_batch(order, pts).cpu().numpy()
            finally:
                ctx.set_policy()
            assert a.shape == b.shape
            if fused is not None:
                assert np.array_equal(fused.cpu().numpy(), a)
            axes = tuple(range(2, a.ndim))
            err = (np.abs(a - b).max(axis=axes) / np.maximum(1.0, np.abs(b).max(axis=axes))).max()
            assert err <= 1e-11, (fa, ka, fb, kb, order, nreq, npts, err)
    # the registered shapes do run fused
    assert runtime.prism_tabulate_batch(prism[0], prism[1], 1, np.zeros((4, 6, 3)) + 0.25) is not None or (ka == 3 and kb == 3)


def test_launches_capture_into_a_hip_graph():
    """After one warm-up call per (element, order, shape) the launch path allocates and synchronises nothing: a sequence of
    tabulations on a side stream captures into a HIP graph (torch.cuda.CUDAGraph) and replays bit-equal (INTEGRATION.md 3;
    tools/graph_probe.py: 60 small launches 0.68 ms direct, 0.48 ms replayed)."""
    import torch
    import fiat_amd
    rng = np.random.default_rng(3)
    work = []
    for fam, sd, deg, npts in (("Lagrange", 3, 3, 23), ("Lagrange", 2, 1, 3), ("Nedelec", 3, 1, 4), ("RaviartThomas", 3, 2, 23),
                               ("DiscontinuousLagrange", 3, 4, 23)):
        el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
        e = rng.exponential(size=(300, npts, sd + 1))
        pts = torch.as_tensor((e / e.sum(-1, keepdims=True))[..., 1:].copy()).cuda()
        out = torch.empty(el.device_polyset().out_shape(1, 300, npts), dtype=torch.float64, device="cuda")
        work.append((el, pts, out))

    def run(stream):
        for el, pts, out in work:
            el.tabulate_batch(1, pts, out=out, stream=stream)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(s)                                             # warm-up: derivative matrices, occupancy queries
    torch.cuda.synchronize()
    ref = [o.clone() for _, _, o in work]
    for _, _, o in work:
        o.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run(s)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for a, (_, _, o) in zip(ref, work):
        assert torch.equal(a, o)


@pytest.mark.parametrize("family,sd,deg,npts", [("Lagrange", 3, 4, 23), ("Lagrange", 3, 4, 17), ("RaviartThomas", 3, 2, 11),
                                                 ("RaviartThomas", 3, 2, 23), ("Nedelec", 3, 3, 23), ("Lagrange", 2, 5, 25)])
def test_odd_request_sizes_on_the_stacked_kernel(family, sd, deg, npts):
    """Requests of an odd number of doubles (odd rows x odd points) on the 8-byte-flush instances of the stacked-matrix kernel,
    against the generic kernel (itself checked against the oracle throughout tests/test_gpu_parity.py): orders 0-2, batches of
    1, 2, 3 (odd requests start on 8-byte boundaries only) and 257 requests."""
    import fiat_amd
    from fiat_amd import runtime
    ctx = runtime.Context.get()
    el = getattr(fiat_amd, family)(fiat_amd.ufc_simplex(sd), deg)
    rows = el.space_dimension() * int(np.prod(el.value_shape() or (1,)))
    assert (rows * npts) % 2 == 1
    rng = np.random.default_rng(31)
    ref_cell = np.array(fiat_amd.ufc_simplex(sd).get_vertices(), dtype=float)
    used = set()
    for order in (0, 1, 2):
        for nreq in (1, 2, 3, 257):
            e = rng.exponential(size=(nreq, npts, sd + 1))
            pts = np.einsum("rpv,vd->rpd", e / e.sum(-1, keepdims=True), ref_cell)
            # (round 4 moved P5 triangles at 25 points to the lane-local / request-per-workgroup kernels by default: the 8-byte
            # instances of the stacked kernel stay reachable -- and tested here -- behind these two policies)
            ctx.set_policy("no_wg", "no_small_values")
            used.add(el.device_polyset().kernel_name(order, nreq, npts))
            a = el.tabulate_batch(order, pts).cpu().numpy()
            ctx.set_policy("no_fixed", "no_small", "no_stacked", "no_coop")
            try:
                b = el.tabulate_batch(order, pts).cpu().numpy()
            finally:
                ctx.set_policy()
            axes = tuple(range(2, a.ndim))
            err = (np.abs(a - b).max(axis=axes) / np.maximum(1.0, np.abs(b).max(axis=axes))).max()
            assert err <= 1e-10, (family, sd, deg, npts, order, nreq, err)
    assert "fxk::tabulate_simplex_stacked" in used
    ctx.check()
