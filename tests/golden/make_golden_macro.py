"""Golden vectors for macro cells / macro elements, from the *unmodified reference*
(build container only; /root/reference does not travel).

    cd /root/repo && \
    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tests/golden/make_golden_macro.py

Writes tests/golden/macro.npz (plain numbers: vertices, topologies, cell-node maps,
seeded points and the reference's tables/coefficients).  Covered reference code:
FIAT/macro.py:83-379 (splits), FIAT/expansions.py:449-490 (macro scatter), :744-811
(cell-node map, point binning), FIAT/lagrange.py:75-88 and
FIAT/discontinuous_lagrange.py:225-241 with a splitting in the variant.
"""
import os

import numpy as np

from FIAT import DiscontinuousLagrange, Lagrange, expansions, ufc_simplex
from FIAT.macro import (AlfeldSplit, IsoSplit, PowellSabin12Split, PowellSabinSplit,
                        WorseyFarinSplit)
from FIAT.polynomial_set import mis

HERE = os.path.dirname(os.path.abspath(__file__))


def jet_keys(sd, order):
    return [a for k in range(order + 1) for a in mis(sd, k)]


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for a in jet_keys(sd, order)])


def rand_points(rng, sd, n):
    e = rng.exponential(size=(n, sd + 1))
    return (e / e.sum(axis=1, keepdims=True))[:, 1:].copy()


SPLITS = {
    "iso_tri": lambda: IsoSplit(ufc_simplex(2)),
    "iso3_tri": lambda: IsoSplit(ufc_simplex(2), 3, "equispaced"),
    "alfeld_tri": lambda: AlfeldSplit(ufc_simplex(2)),
    "ps_tri": lambda: PowellSabinSplit(ufc_simplex(2)),
    "ps12_tri": lambda: PowellSabin12Split(ufc_simplex(2)),
    "iso_tet": lambda: IsoSplit(ufc_simplex(3)),
    "alfeld_tet": lambda: AlfeldSplit(ufc_simplex(3)),
    "wf_tet": lambda: WorseyFarinSplit(ufc_simplex(3)),
    "iso_line": lambda: IsoSplit(ufc_simplex(1)),
}


def special_points(S):
    """Points on the interfaces of the complex: its vertices, the midpoints of every edge and the
    barycentres of every facet -- where the binning is not unique."""
    top = S.get_topology()
    V = np.asarray(S.get_vertices())
    pts = [V[list(top[d][e])].mean(axis=0) for d in sorted(top) for e in sorted(top[d])]
    return np.asarray(pts)


def main():
    rng = np.random.default_rng(20261004)
    out = {}
    for name, make in SPLITS.items():
        S = make()
        sd = S.get_spatial_dimension()
        top = S.get_topology()
        out[f"{name}/verts"] = np.asarray(S.get_vertices())
        for d in sorted(top):
            out[f"{name}/top{d}"] = np.asarray([top[d][e] for e in sorted(top[d])], dtype=np.int64)
        c2p = S.get_child_to_parent()
        for d in sorted(top):
            out[f"{name}/c2p{d}"] = np.asarray([c2p[d][e] for e in sorted(top[d])], dtype=np.int64)
        conn = S.get_cell_connectivity()
        for d in sorted(top):
            out[f"{name}/conn{d}"] = np.asarray([conn[c][d] for c in sorted(top[sd])], dtype=np.int64)
        for d in sorted(top):
            out[f"{name}/interior{d}"] = np.asarray(S.get_interior_facets(d), dtype=np.int64)
        pts = np.vstack([rand_points(rng, sd, 17), special_points(S)])
        out[f"{name}/pts"] = pts
        nmax = 3 if sd < 3 or name != "wf_tet" else 2
        for variant in (None, "bubble"):
            U = expansions.ExpansionSet(S, variant=variant)
            vn = variant or "none"
            for n in range(0 if variant is None else 1, nmax + 1):
                out[f"{name}/{vn}/n{n}/cell_node_map"] = np.asarray(U.get_cell_node_map(n), dtype=np.int64)
                out[f"{name}/{vn}/n{n}/num_members"] = np.int64(U.get_num_members(n))
                for order in (0, 2):
                    out[f"{name}/{vn}/n{n}/tab{order}"] = stack(U._tabulate(n, pts, order), sd, order)
        # binning on its own (expansions.py:771-811), both modes
        for unique in (True, False):
            cpm = expansions.compute_cell_point_map(S, pts, unique=unique)
            mask = np.zeros((len(top[sd]), len(pts)), dtype=np.int64)
            for c, ipts in cpm.items():
                mask[c, ipts] = 1
            out[f"{name}/bins_unique{int(unique)}"] = mask

    # ---- elements on split cells --------------------------------------------------------
    elements = {
        "cg2_alfeld_tri": lambda: Lagrange(ufc_simplex(2), 2, "equispaced,alfeld"),
        "cg1_iso_tri": lambda: Lagrange(ufc_simplex(2), 1, "equispaced,iso"),
        "cg2_iso_tri": lambda: Lagrange(ufc_simplex(2), 2, "equispaced,iso"),
        "cg1_iso3_tri": lambda: Lagrange(ufc_simplex(2), 1, "equispaced,iso(3)"),
        "cg2_ps_tri": lambda: Lagrange(ufc_simplex(2), 2, "equispaced,powell-sabin"),
        "cg1_iso_tet": lambda: Lagrange(ufc_simplex(3), 1, "equispaced,iso"),
        "cg3_alfeld_tet": lambda: Lagrange(ufc_simplex(3), 3, "equispaced,alfeld"),
        "cg2_wf_tet": lambda: Lagrange(ufc_simplex(3), 2, "equispaced,worsey-farin"),
        "dg2_alfeld_tri": lambda: DiscontinuousLagrange(ufc_simplex(2), 2, "equispaced_interior,alfeld"),
        "dg1_iso_tri": lambda: DiscontinuousLagrange(ufc_simplex(2), 1, "equispaced_interior,iso"),
        "dg1_alfeld_tet": lambda: DiscontinuousLagrange(ufc_simplex(3), 1, "equispaced_interior,alfeld"),
    }
    names = []
    for name, make in elements.items():
        e = make()
        S = e.get_reference_element()
        sd = S.get_spatial_dimension()
        names.append(name)
        out[f"el/{name}/coeffs"] = np.asarray(e.get_coeffs())
        ids = e.entity_dofs()
        flat = [(d, ent, dof) for d in sorted(ids) for ent in sorted(ids[d]) for dof in ids[d][ent]]
        out[f"el/{name}/entity_dofs"] = np.asarray(flat, dtype=np.int64).reshape(-1, 3)
        out[f"el/{name}/nodes"] = np.asarray([list(ell.get_point_dict().keys())[0] for ell in e.dual_basis()])
        pts = np.vstack([rand_points(rng, sd, 23), special_points(S)[::3]])
        out[f"el/{name}/pts"] = pts
        out[f"el/{name}/tab1"] = stack(e.tabulate(1, pts), sd, 1)
        out[f"el/{name}/tab2"] = stack(e.tabulate(2, pts), sd, 2)
    out["el/names"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, "macro.npz"), **out)
    print("macro.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
