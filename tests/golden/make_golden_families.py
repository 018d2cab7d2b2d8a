"""Golden vectors for the wider-families row, generated from the
*unmodified reference* in the build container:

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tests/golden/make_golden_families.py

Brezzi-Douglas-Marini, second-kind Nedelec, Regge, Hellan-Herrmann-Johnson, point variants of Nedelec / Raviart-Thomas,
cubic Hermite and Morley elements on the UFC interval / triangle / tetrahedron:
nodal coefficients, entity -> dof map, the tables of tabulate(1, points) at seeded points.
Plain numbers only."""
import json
import os

import numpy as np

from FIAT import (BrezziDouglasMarini, CubicHermite, HellanHerrmannJohnson, Morley, Nedelec, NedelecSecondKind, RaviartThomas,
                  Regge, ufc_simplex)
from FIAT.polynomial_set import mis

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("bdm", BrezziDouglasMarini, 2, (1, 2, 3)), ("bdm", BrezziDouglasMarini, 3, (1, 2)),
         ("n2curl", NedelecSecondKind, 2, (1, 2, 3)), ("n2curl", NedelecSecondKind, 3, (1, 2)),
         # derivative functionals in the dual set (FIAT/dual_set.py:175-205)
         ("hermite", CubicHermite, 1, (3,)), ("hermite", CubicHermite, 2, (3,)), ("hermite", CubicHermite, 3, (3,)),
         ("morley", Morley, 2, (2,)), ("morley", Morley, 3, (2,)),
         # point variants (normal / tangential point evaluations)
         ("rtpt", lambda c, k: RaviartThomas(c, k, variant="point"), 2, (1, 2)),
         ("rtpt", lambda c, k: RaviartThomas(c, k, variant="point"), 3, (1, 2)),
         ("nedpt", lambda c, k: Nedelec(c, k, variant="point"), 2, (1, 2)),
         ("nedpt", lambda c, k: Nedelec(c, k, variant="point"), 3, (1, 2)),
         ("bdmpt", lambda c, k: BrezziDouglasMarini(c, k, variant="point"), 2, (1, 2)),
         ("bdmpt", lambda c, k: BrezziDouglasMarini(c, k, variant="point"), 3, (1, 2)),
         ("n2curlpt", lambda c, k: NedelecSecondKind(c, k, variant="point"), 2, (1, 2)),
         ("n2curlpt", lambda c, k: NedelecSecondKind(c, k, variant="point"), 3, (1, 2)),
         # symmetric-matrix-valued elements (ONSymTensorPolynomialSet)
         ("regge", Regge, 2, (0, 1, 2)), ("regge", Regge, 3, (0, 1)),
         ("reggept", lambda c, k: Regge(c, k, variant="point"), 2, (1,)), ("reggept", lambda c, k: Regge(c, k, variant="point"), 3, (1,)),
         ("hhj", HellanHerrmannJohnson, 2, (0, 1, 2)), ("hhj", HellanHerrmannJohnson, 3, (0, 1)),
         ("hhjpt", lambda c, k: HellanHerrmannJohnson(c, k, variant="point"), 3, (1,))]


def main():
    rng = np.random.default_rng(424)
    out = {}
    for sd in (2, 3, 1):
        e = rng.exponential(size=(6, sd + 1))
        out[f"pts_sd{sd}"] = (e / e.sum(axis=1, keepdims=True))[:, 1:]
    for name, cls, sd, degrees in CASES:
        for k in degrees:
            el = cls(ufc_simplex(sd), k)
            key = f"{name}{k}_sd{sd}"
            out[key + "_coeffs"] = np.asarray(el.get_coeffs())
            tab = el.tabulate(1, out[f"pts_sd{sd}"])
            out[key + "_tab"] = np.stack([np.asarray(tab[a]) for j in range(2) for a in mis(sd, j)])
            out[key + "_entity_dofs"] = np.array(json.dumps({str(d): {str(i): [int(x) for x in v] for i, v in ents.items()}
                                                             for d, ents in el.entity_dofs().items()}))
            out[key + "_mapping"] = np.array(el.mapping()[0])
    # matrix-valued elements built DIRECTLY on physical cells (double Piola push-forward check)
    from FIAT import reference_element
    for sd in (2, 3):
        base = ufc_simplex(sd)
        ref = np.array(base.get_vertices(), dtype=float)
        ncell = 3
        A = np.eye(sd) + 0.3 * rng.standard_normal((ncell, sd, sd))
        for i in range(ncell):
            if np.linalg.det(A[i]) < 0:
                A[i, :, 0] *= -1
        A[ncell - 1, :, 0] *= -1                    # one negatively oriented cell
        verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((ncell, 1, sd))
        e = rng.exponential(size=(ncell, 5, sd + 1))
        pts = np.einsum("rpv,rvd->rpd", e / e.sum(axis=-1, keepdims=True), verts)
        out[f"phys_verts_sd{sd}"] = verts
        out[f"phys_pts_sd{sd}"] = pts
        for name, cls in (("regge", Regge), ("hhj", HellanHerrmannJohnson)):
            tabs = []
            for i in range(ncell):
                cell = reference_element.UFCSimplex(base.get_shape(), tuple(map(tuple, verts[i])), base.get_topology())
                tab = cls(cell, 1).tabulate(1, pts[i])
                tabs.append(np.stack([np.asarray(tab[a]) for j in range(2) for a in mis(sd, j)]))
            out[f"{name}1_phys_sd{sd}_tab"] = np.stack(tabs)
    np.savez_compressed(os.path.join(HERE, "families.npz"), **out)
    print("wrote families.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
