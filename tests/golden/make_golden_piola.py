"""Golden vectors for the push-forward row, generated from the
*unmodified reference* in the build container:

    \
    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tests/golden/make_golden_piola.py

For random non-degenerate physical simplices K: the reference's Nedelec / Raviart-Thomas /
Lagrange element constructed DIRECTLY on K, tabulated (order 1) at points of K.  The GPU
path must reproduce these tables from the reference-cell coefficients plus the cell
geometry (covariant / contravariant Piola, affine pull-back).  Plain numbers only."""
import os

import numpy as np

from FIAT import Lagrange, Nedelec, RaviartThomas, reference_element, ufc_simplex
from FIAT.polynomial_set import mis

HERE = os.path.dirname(os.path.abspath(__file__))


def physical_cell(sd, verts):
    base = ufc_simplex(sd)
    return reference_element.UFCSimplex(base.get_shape(), tuple(map(tuple, verts)), base.get_topology())


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for k in range(order + 1) for a in mis(sd, k)])


def main():
    rng = np.random.default_rng(8126)
    out = {}
    for sd in (2, 3):
        ref = np.array(ufc_simplex(sd).get_vertices(), dtype=float)
        ncell = 4
        A = np.eye(sd) + 0.3 * rng.standard_normal((ncell, sd, sd))
        for i in range(ncell):                      # keep the orientation (positive Jacobian) for half, flip for one
            if np.linalg.det(A[i]) < 0:
                A[i, :, 0] *= -1
        A[ncell - 1, :, 0] *= -1                    # one negatively oriented cell
        b = rng.standard_normal((ncell, 1, sd))
        verts = np.einsum("vd,red->rve", ref, A) + b
        e = rng.exponential(size=(ncell, 7, sd + 1))
        bary = e / e.sum(axis=-1, keepdims=True)
        pts = np.einsum("rpv,rvd->rpd", bary, verts)
        out[f"verts_sd{sd}"] = verts
        out[f"pts_sd{sd}"] = pts
        for name, cls, deg in (("n1", Nedelec, 1), ("n2", Nedelec, 2), ("rt1", RaviartThomas, 1), ("rt2", RaviartThomas, 2),
                               ("p2", Lagrange, 2)):
            tabs = []
            for i in range(ncell):
                el = cls(physical_cell(sd, verts[i]), deg)
                tabs.append(stack(el.tabulate(1, pts[i]), sd, 1))
            out[f"{name}_sd{sd}_tab"] = np.stack(tabs)
            out[f"{name}_sd{sd}_refcoeffs"] = np.asarray(cls(ufc_simplex(sd), deg).get_coeffs())
    np.savez_compressed(os.path.join(HERE, "piola.npz"), **out)
    print("wrote piola.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
