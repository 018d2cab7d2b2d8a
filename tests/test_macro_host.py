"""Macro cells on the CPU: the splits of fiat_amd/macro.py (vertices, topology, parent/child relations,
cell connectivity, cell-node maps) and the oracle's macro tabulation against golden vectors produced
by the reference itself (tests/golden/make_golden_macro.py; FIAT/macro.py:83-379,
FIAT/expansions.py:449-490,744-811)."""
import numpy as np
import pytest

from oracle import fiat_oracle as fo

SPLITS = ["iso_tri", "iso3_tri", "alfeld_tri", "ps_tri", "ps12_tri", "iso_tet", "alfeld_tet", "wf_tet", "iso_line"]


def make_split(name):
    from fiat_amd import macro, ufc_simplex
    kind, cell = name.split("_")
    T = ufc_simplex({"line": 1, "tri": 2, "tet": 3}[cell])
    return {"iso": lambda: macro.IsoSplit(T), "iso3": lambda: macro.IsoSplit(T, 3, "equispaced"),
            "alfeld": lambda: macro.AlfeldSplit(T), "ps": lambda: macro.PowellSabinSplit(T),
            "ps12": lambda: macro.PowellSabin12Split(T), "wf": lambda: macro.WorseyFarinSplit(T)}[kind]()


@pytest.mark.parametrize("name", SPLITS)
def test_split_topology(golden, name):
    G = golden("macro")
    S = make_split(name)
    sd = S.get_spatial_dimension()
    top = S.get_topology()
    assert S.is_macrocell() and S.get_parent().get_spatial_dimension() == sd
    np.testing.assert_allclose(np.array(S.get_vertices()), G[f"{name}/verts"], atol=1e-15)
    c2p, conn = S.get_child_to_parent(), S.get_cell_connectivity()
    for d in top:
        assert np.array_equal(np.array([top[d][e] for e in sorted(top[d])]), G[f"{name}/top{d}"])
        assert np.array_equal(np.array([c2p[d][e] for e in sorted(top[d])]), G[f"{name}/c2p{d}"])
        assert np.array_equal(np.array([conn[c][d] for c in sorted(top[sd])]), G[f"{name}/conn{d}"])
        assert list(S.get_interior_facets(d)) == list(G[f"{name}/interior{d}"])
    # every parent entity is covered by its children
    p2c = S.get_parent_to_children()
    assert sum(len(v) for d in p2c for v in p2c[d].values()) == sum(len(top[d]) for d in top)
    assert abs(S.volume() - S.get_parent().volume()) < 1e-14


@pytest.mark.parametrize("name", SPLITS)
def test_cell_node_map(golden, name):
    from fiat_amd import expansions
    G = golden("macro")
    S = make_split(name)
    for vn, cont in (("none", None), ("bubble", "C0")):
        for n in range(4):
            key = f"{name}/{vn}/n{n}/cell_node_map"
            if key not in G.files:
                continue
            assert np.array_equal(expansions.polynomial_cell_node_map(S, n, cont), G[key])
            assert expansions.polynomial_dimension(S, n, cont) == int(G[f"{name}/{vn}/n{n}/num_members"])


@pytest.mark.parametrize("name", SPLITS)
def test_oracle_macro_tabulate(golden, name):
    """The NumPy restatement of the macro scatter equals the reference (values, gradients, Hessians),
    also at points on the interfaces of the complex (non-unique binning)."""
    G = golden("macro")
    V = G[f"{name}/verts"]
    sd = V.shape[1]
    cells = [V[list(c)] for c in G[f"{name}/top{sd}"]]
    parent = np.array(fo.UFC_SIMPLEX[sd])
    pts = G[f"{name}/pts"]
    for u in (0, 1):
        mask = fo.macro_cell_point_map(parent, cells, pts, unique=bool(u))
        assert np.array_equal(mask, G[f"{name}/bins_unique{u}"].astype(bool))
    for vn in ("none", "bubble"):
        for n in range(4):
            key = f"{name}/{vn}/n{n}/cell_node_map"
            if key not in G.files:
                continue
            nm = int(G[f"{name}/{vn}/n{n}/num_members"])
            for order in (0, 2):
                ref = G[f"{name}/{vn}/n{n}/tab{order}"]
                t = fo.macro_expansion_tabulate(parent, cells, G[key], nm, n, pts, order, None,
                                                None if vn == "none" else "bubble")
                got = np.stack([t[a] for a in fo.jet_indices(sd, order)])
                assert np.max(np.abs(got - ref)) <= 5e-13 * max(1.0, np.max(np.abs(ref)))


def test_variant_parsing():
    from fiat_amd import macro
    from fiat_amd.check_format_variant import parse_lagrange_variant
    assert parse_lagrange_variant("equispaced,alfeld") == (macro.AlfeldSplit, "equispaced")
    assert parse_lagrange_variant("equispaced,iso")[0] is macro.IsoSplit
    split, pv = parse_lagrange_variant("equispaced,iso(3)")
    from fiat_amd import ufc_simplex
    S = split(ufc_simplex(2))
    assert isinstance(S, macro.IsoSplit) and S.degree == 3 and len(S.get_topology()[2]) == 9
    with pytest.raises(ValueError):     # DG nodes on sub-cell boundaries (check_format_variant.py:93-94)
        parse_lagrange_variant("equispaced,alfeld", discontinuous=True)
    with pytest.raises(ValueError):
        parse_lagrange_variant("equispaced,isotropic")
