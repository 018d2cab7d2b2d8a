"""Build-container only: writes fiat_amd/data/simplex_rules.npz, the point/weight tables behind
``create_quadrature(cell, degree, "default")`` on triangles and tetrahedra.

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tools/make_rule_tables.py

The reference's default scheme is DATA, not an algorithm: Xiao-Gimbutas rules (H. Xiao, Z. Gimbutas, Comput. Math.
Appl. 59 (2010) 663-676) tabulated on the symmetric simplex for triangle degrees 1..50 and tetrahedron degrees
1..15 (FIAT/xg_quad_data.py, selected by FIAT/quadrature_schemes.py:324-353), and the classical Zienkiewicz-Taylor /
Strang-Fix rules for the lowest degrees, given on the UFC simplex (:356-419).  This script imports the unmodified
reference, reads those numbers and stores them as plain arrays (per dimension: the degrees, an offset table, points,
weights, and the vertices of the cell the points live on).  No reference code is copied; the product maps the
tables to the requested cell itself (fiat_amd/quadrature.py)."""
import os

import numpy as np

from FIAT import xg_quad_data
from FIAT.quadrature_schemes import create_quadrature
from FIAT.reference_element import symmetric_simplex, ufc_simplex

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "fiat_amd", "data", "simplex_rules.npz")


def pack(rules):
    """{degree: (pts, wts)} -> degrees, offsets, pts, wts (concatenated)."""
    degrees = np.array(sorted(rules), dtype=np.int32)
    counts = [len(rules[d][1]) for d in degrees]
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pts = np.concatenate([np.asarray(rules[d][0], dtype=np.float64) for d in degrees])
    wts = np.concatenate([np.asarray(rules[d][1], dtype=np.float64) for d in degrees])
    return degrees, offsets, pts, wts


def main():
    out = {}
    low = {2: (0, 1, 2, 3), 3: (0, 1, 2)}       # degrees served by the classical rules on the UFC simplex
    for dim, table in ((2, xg_quad_data.triangle_table), (3, xg_quad_data.tetrahedron_table)):
        xg = {d: (np.asarray(v["points"], dtype=np.float64).reshape(-1, dim), np.asarray(v["weights"], dtype=np.float64))
              for d, v in table.items() if d not in low[dim]}
        classical = {}
        for d in low[dim]:
            Q = create_quadrature(ufc_simplex(dim), d)
            classical[d] = (Q.get_points(), Q.get_weights())
        for name, rules, cell in (("xg", xg, symmetric_simplex(dim)), ("classical", classical, ufc_simplex(dim))):
            deg, off, pts, wts = pack(rules)
            out[f"{name}{dim}_degrees"], out[f"{name}{dim}_offsets"] = deg, off
            out[f"{name}{dim}_points"], out[f"{name}{dim}_weights"] = pts, wts
            out[f"{name}{dim}_cell"] = np.array(cell.get_vertices(), dtype=np.float64)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()}, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
