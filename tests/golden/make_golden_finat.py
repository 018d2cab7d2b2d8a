"""Golden vectors for the FInAT-side adapter (SURVEY.md 8f rank 3 / rank 4 ``dual_basis``), generated from the
*unmodified reference* in the build container:

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 python -B tests/golden/make_golden_finat.py

``finat/__init__.py`` imports ``finat.ufl`` (needs the absent ``ufl`` package); none of the modules on this path
does.  A bare package object is registered under the name ``finat`` so that ``__init__`` never runs, and
``finat.fiat_elements``, ``finat.point_set``, ``finat.runtime_tabulated``, ``finat.tensor_product`` are imported
unmodified from /root/reference/finat.  GEM expressions are turned into arrays with ``gem.interpreter.evaluate``.

What is stored (plain numbers and strings):
* ``be_<case>_*``: ``FiatElement.basis_evaluation(order, PointSet)`` (finat/fiat_elements.py:60-123): the evaluated table
  of every multi-index with the point axis LAST, and ``kind`` = 0 pointwise / 1 cellwise constant (no point index in
  the expression) / 2 zero (no point index and the reference's own ``numpy.zeros``);
* ``db_<case>_*``: ``_dual_basis`` (finat/fiat_elements.py:163-262): dense Q, the unique points, whether the reference
  expressed Q symbolically as a Kronecker delta;
* ``rt_*``: the ``gem.Variable`` names and shapes ``RuntimeTabulated.basis_evaluation`` asks for
  (finat/runtime_tabulated.py:68-95);
* ``tp_*``: ``TensorProductElement.basis_evaluation`` on a tensor point set (finat/tensor_product.py:98-144): product
  tables per multi-index (axes: basis indices of the factors, then the point indices of the factors; a factor whose table the
  reference made cellwise constant contributes an axis of extent 1);
* ``esd_*``: ``entity_support_dofs`` of FIAT (FIAT/finite_element.py:222-264) and of FInAT
  (finat/finiteelementbase.py:85-119) as flat integer arrays.
"""
import os
import sys
import types

import numpy as np

pkg = types.ModuleType("finat")
pkg.__path__ = ["/root/reference/finat"]
sys.modules["finat"] = pkg

import FIAT  # noqa: E402
import gem  # noqa: E402
from FIAT.finite_element import entity_support_dofs as fiat_esd  # noqa: E402
from FIAT.polynomial_set import mis  # noqa: E402
from gem.interpreter import evaluate  # noqa: E402

import finat.fiat_elements as ffe  # noqa: E402
from finat.point_set import PointSet, TensorPointSet  # noqa: E402
from finat.runtime_tabulated import RuntimeTabulated  # noqa: E402
from finat.tensor_product import TensorProductElement  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def simplex_points(rng, sd, n):
    e = rng.exponential(size=(n, sd + 1))
    return (e / e.sum(axis=1, keepdims=True))[:, 1:].copy()


def dense(expr, point_indices):
    """Evaluate ``expr`` and return (array with the free point indices moved to the END in the order of
    ``point_indices``, mask of the point indices present)."""
    res, = evaluate([expr])
    arr = np.asarray(res.arr)
    fids = list(res.fids)
    present = [p in fids for p in point_indices]
    # evaluate() puts the free indices first, then the shape
    order = [fids.index(p) for p in point_indices if p in fids]
    assert sorted(order) == list(range(len(fids))), "unexpected free index"
    nf = len(fids)
    arr = np.transpose(arr, list(range(nf, arr.ndim)) + order)
    return arr, present


def variables(expr):
    found, stack, seen = [], [expr], set()
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if isinstance(n, gem.Variable):
            found.append(n)
        stack.extend(n.children)
    return found


BASIS_CASES = [
    # name, finat class, sd, degree, kwargs, order, npts
    ("P2tri", "Lagrange", 2, 2, {}, 3, 7),
    ("P3tet", "Lagrange", 3, 3, {}, 1, 23),
    ("P1tet", "Lagrange", 3, 1, {}, 2, 5),
    ("DG1tet", "DiscontinuousLagrange", 3, 1, {}, 2, 6),
    ("N2tet", "Nedelec", 3, 2, {}, 2, 9),
    ("RT2tet", "RaviartThomas", 3, 2, {}, 1, 23),
    ("RT1tri", "RaviartThomas", 2, 1, {}, 2, 4),
    ("N2ndtet", "NedelecSecondKind", 3, 1, {}, 2, 5),
    ("Regge1tri", "Regge", 2, 1, {}, 2, 6),
    ("P1iso2tri", "Lagrange", 2, 1, {"variant": "equispaced,iso(2)"}, 1, 12),
    ("P1alfeldtri", "Lagrange", 2, 1, {"variant": "equispaced,alfeld"}, 1, 12),
]

DUAL_CASES = [
    ("P2tri", "Lagrange", 2, 2), ("P3tet", "Lagrange", 3, 3), ("DG2tet", "DiscontinuousLagrange", 3, 2),
    ("RT2tri", "RaviartThomas", 2, 2), ("N2tet", "Nedelec", 3, 2), ("BDM1tet", "BrezziDouglasMarini", 3, 1),
    ("Regge1tri", "Regge", 2, 1), ("RT2tet", "RaviartThomas", 3, 2),
]

ESD_CASES = [
    ("P3tet", "Lagrange", 3, 3, {}), ("N2tet", "Nedelec", 3, 2, {}), ("RT2tet", "RaviartThomas", 3, 2, {}),
    ("DG2tet", "DiscontinuousLagrange", 3, 2, {}), ("P2tri", "Lagrange", 2, 2, {}), ("RT1tri", "RaviartThomas", 2, 1, {}),
    ("BDM1tet", "BrezziDouglasMarini", 3, 1, {}), ("P1int", "Lagrange", 1, 1, {}),
]


def main():
    rng = np.random.default_rng(33)
    out = {}
    # ---- basis_evaluation ---------------------------------------------------------------------------
    out["be_cases"] = np.array([c[0] for c in BASIS_CASES])
    for name, cls, sd, degree, kw, order, npts in BASIS_CASES:
        el = getattr(ffe, cls)(FIAT.ufc_simplex(sd), degree, **kw)
        pts = simplex_points(rng, sd, npts)
        ps = PointSet(pts)
        res = el.basis_evaluation(order, ps)
        alphas = [a for k in range(order + 1) for a in mis(sd, k)]
        assert list(res) == alphas
        out[f"be_{name}_pts"] = pts
        out[f"be_{name}_meta"] = np.array([sd, degree, order, el.degree, int(el.complex.is_simplex())])
        kinds = []
        for t, alpha in enumerate(alphas):
            arr, present = dense(res[alpha], ps.indices)
            if not present[0] and not arr.any():
                kind = 2              # the reference's own numpy.zeros (:104-108); a gem.Literal like the constant tables
                assert sum(alpha) > el.degree
            elif not present[0]:
                kind = 1
            else:
                kind = 0
            kinds.append(kind)
            out[f"be_{name}_t{t}"] = arr
        out[f"be_{name}_kinds"] = np.array(kinds)
        out[f"be_{name}_shape"] = np.array(el.index_shape + el.value_shape)
    # a facet: entity=(dim, id)
    el = ffe.Lagrange(FIAT.ufc_simplex(3), 2)
    fpts = simplex_points(rng, 2, 5)
    ps = PointSet(fpts)
    res = el.basis_evaluation(1, ps, entity=(2, 1))
    out["be_facet_pts"] = fpts
    for t, alpha in enumerate([a for k in range(2) for a in mis(3, k)]):
        out[f"be_facet_t{t}"] = dense(res[alpha], ps.indices)[0]
    # ---- dual_basis ----------------------------------------------------------------------------------
    out["db_cases"] = np.array([c[0] for c in DUAL_CASES])
    for name, cls, sd, degree in DUAL_CASES:
        el = getattr(ffe, cls)(FIAT.ufc_simplex(sd), degree)
        Q, pts = el._dual_basis
        out[f"db_{name}_identity"] = np.array(int(not isinstance(Q, gem.Literal)))
        arr, _ = dense(Q, ())
        out[f"db_{name}_Q"] = arr
        out[f"db_{name}_pts"] = np.asarray(pts)
        # the public property: same numbers behind a fresh point index
        Qx, x = el.dual_basis
        arrx, present = dense(Qx, x.indices)
        assert present == [True]
        np.testing.assert_array_equal(np.moveaxis(arrx, -1, 1), arr)
        np.testing.assert_array_equal(x.points, pts)
    try:
        ffe.FiatElement(FIAT.CubicHermite(FIAT.ufc_simplex(2))).dual_basis
        out["db_hermite_raises"] = np.array(0)
    except NotImplementedError:
        out["db_hermite_raises"] = np.array(1)
    # ---- RuntimeTabulated ----------------------------------------------------------------------------
    cell = FIAT.ufc_simplex(1)
    rt_names, rt_shapes = [], []
    for degree, kw, order, npts in ((3, dict(variant="equispaced", shift_axes=1, restriction='+', continuous=True), 2, 5),
                                    (2, dict(variant="gll", shift_axes=0, restriction=None, continuous=False), 1, 4),
                                    (1, dict(variant="spectral", shift_axes=2, restriction='-', continuous=True), 0, 3)):
        rt = RuntimeTabulated(cell, degree, **kw)
        ps = PointSet(np.linspace(0.1, 0.9, npts)[:, None])
        res = rt.basis_evaluation(order, ps)
        for alpha, expr in res.items():
            v, = variables(expr)
            rt_names.append(v.name)
            rt_shapes.append(list(v.shape) + [rt.formdegree, rt.space_dimension()])
            assert expr.shape == rt.index_shape and set(expr.free_indices) == set(ps.indices)
    out["rt_names"] = np.array(rt_names)
    out["rt_shapes"] = np.array(rt_shapes)
    # ---- tensor product ------------------------------------------------------------------------------
    factors = [ffe.Lagrange(cell, 2), ffe.Lagrange(cell, 3), ffe.DiscontinuousLagrange(cell, 1)]
    tp = TensorProductElement(factors)
    coords = [np.linspace(0.05, 0.95, n)[:, None] for n in (4, 5, 3)]
    tps = TensorPointSet([PointSet(c) for c in coords])
    res = tp.basis_evaluation(1, tps)
    out["tp_index_shape"] = np.array(tp.index_shape)
    for i, c in enumerate(coords):
        out[f"tp_coords{i}"] = c
    deltas = [a for k in range(2) for a in mis(3, k)]
    assert list(res) == deltas
    out["tp_deltas"] = np.array(deltas)
    for t, Delta in enumerate(deltas):
        arr, present = dense(res[Delta], tps.indices)
        # re-insert the point axes the reference dropped (cellwise-constant factor tables) with extent 1
        shape = list(arr.shape[:len(tp.index_shape)])
        k = len(tp.index_shape)
        for pr in present:
            if pr:
                shape.append(arr.shape[k])
                k += 1
            else:
                shape.append(1)
        out[f"tp_t{t}"] = arr.reshape(shape)
        out[f"tp_present{t}"] = np.array(present, dtype=np.int64)
    # ---- entity_support_dofs -------------------------------------------------------------------------
    out["esd_cases"] = np.array([c[0] for c in ESD_CASES] + ["P2xP1prism"])

    def put_esd(prefix, dims, getter):
        keys = []
        for dim in dims:
            d = getter(dim)
            dtag = "-".join(map(str, dim)) if isinstance(dim, tuple) else str(dim)
            for f, dofs in d.items():
                out[f"{prefix}_{dtag}_{f}"] = np.array(dofs, dtype=np.int64)
                keys.append(f"{dtag}_{f}")
        out[f"{prefix}_keys"] = np.array(keys)

    for name, cls, sd, degree, kw in ESD_CASES:
        fiat_el = getattr(FIAT, cls)(FIAT.ufc_simplex(sd), degree, **kw)
        put_esd(f"esd_fiat_{name}", range(sd + 1), lambda dim: fiat_esd(fiat_el, dim))
        finat_el = getattr(ffe, cls)(FIAT.ufc_simplex(sd), degree, **kw)
        put_esd(f"esd_finat_{name}", range(sd + 1), lambda dim: finat_el.entity_support_dofs()[dim])
    prism = FIAT.TensorProductElement(FIAT.Lagrange(FIAT.ufc_simplex(2), 2), FIAT.Lagrange(FIAT.ufc_simplex(1), 1))
    out["esd_prism_degree"] = np.array(prism.degree())
    put_esd("esd_fiat_P2xP1prism", sorted(prism.entity_dofs()), lambda dim: fiat_esd(prism, dim))

    path = os.path.join(HERE, "finat.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
