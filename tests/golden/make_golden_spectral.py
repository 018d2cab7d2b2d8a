"""Golden vectors for the spectral point families (GLL / GL / Chebyshev), from the unmodified reference with
oracle/restated_deps standing in for ``recursivenodes`` (build container only).

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 \
        python -B tests/golden/make_golden_spectral.py

On intervals the lattices are the 1-D node families themselves (Gauss-Lobatto-Legendre = end points + roots of
P_n', Gauss-Legendre = roots of P_{n+1}: defined mathematically, checked against closed forms in the tests).  On
triangles and tetrahedra they come from the recursive rule of ``recursivenodes``, which is absent from the image:
those fixtures pin this repository's host code against the reference code paths *given* our restatement of that
rule (parity unpinned with respect to the third-party package itself).
"""
import os

import numpy as np

from FIAT import (DiscontinuousLagrange, GaussLegendre, GaussLobattoLegendre, Lagrange, TensorProductElement,
                  ufc_simplex)
from FIAT.polynomial_set import mis
from FIAT.reference_element import make_lattice

HERE = os.path.dirname(os.path.abspath(__file__))


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for k in range(order + 1) for a in mis(sd, k)])


def main():
    rng = np.random.default_rng(20261006)
    out = {}
    for sd in (1, 2, 3):
        V = ufc_simplex(sd).get_vertices()
        for variant in ("gll", "gl", "lgc", "gc", "equispaced_interior"):
            for n in (1, 2, 3, 4, 5):
                out[f"lattice/{variant}/sd{sd}/n{n}"] = np.asarray(make_lattice(V, n, variant=variant))
            out[f"lattice/{variant}/sd{sd}/n4_int1"] = np.asarray(make_lattice(V, 4, 1, variant=variant)).reshape(-1, sd)
    elements = {
        "gll_line4": lambda: GaussLobattoLegendre(ufc_simplex(1), 4),
        "gll_tri3": lambda: GaussLobattoLegendre(ufc_simplex(2), 3),
        "gll_tet3": lambda: GaussLobattoLegendre(ufc_simplex(3), 3),
        "cg_spectral_tri4": lambda: Lagrange(ufc_simplex(2), 4, "spectral"),
        "gl_line3": lambda: GaussLegendre(ufc_simplex(1), 3),
        "gl_tri2": lambda: GaussLegendre(ufc_simplex(2), 2),
        "dg_spectral_tet2": lambda: DiscontinuousLagrange(ufc_simplex(3), 2, "spectral"),
        "cg_chebyshev_tri3": lambda: Lagrange(ufc_simplex(2), 3, "chebyshev"),
    }
    for name, make in elements.items():
        e = make()
        sd = e.get_reference_element().get_spatial_dimension()
        out[f"el/{name}/coeffs"] = np.asarray(e.get_coeffs())
        ids = e.entity_dofs()
        flat = [(d, ent, dof) for d in sorted(ids) for ent in sorted(ids[d]) for dof in ids[d][ent]]
        out[f"el/{name}/entity_dofs"] = np.asarray(flat, dtype=np.int64).reshape(-1, 3)
        out[f"el/{name}/nodes"] = np.asarray([list(ell.get_point_dict().keys())[0] for ell in e.dual_basis()])
        ex = rng.exponential(size=(11, sd + 1))
        pts = (ex / ex.sum(1, keepdims=True))[:, 1:].copy()
        out[f"el/{name}/pts"] = pts
        out[f"el/{name}/tab1"] = stack(e.tabulate(1, pts), sd, 1)
    # the spectral-element hexahedron: P4 GLL x P4 GLL x P4 GLL
    A = GaussLobattoLegendre(ufc_simplex(1), 4)
    hexa = TensorProductElement(TensorProductElement(A, A), A)
    pts = rng.uniform(size=(9, 3))
    out["hex_gll4/pts"] = pts
    tab = hexa.tabulate(1, pts)
    out["hex_gll4/tab1"] = np.stack([np.asarray(tab[a]) for a in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]])
    np.savez_compressed(os.path.join(HERE, "spectral.npz"), **out)
    print("spectral.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
