"""Golden vectors of round 4, generated from the *unmodified reference* in the build container:

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 python -B tests/golden/make_golden_round4.py

``ho_*``: derivative orders 5 and 6 (FIAT/expansions.py:66-137 carries the Leibniz rule to any order; above its recurrence
order the reference switches to differentiation matrices, :438-446) -- of elements on the UFC cell at reference points, and
of the same families built by the reference ON physical cells (incl. a negatively oriented one) at physical points: what
``tabulate(order, points)`` and ``tabulate_batch(order, points, verts=cells)`` must return at orders the earlier fixtures
(round2.npz / round3.npz: orders <= 4) do not reach.  Plain numbers only."""
import os

import numpy as np

import FIAT
from FIAT import polynomial_set
from FIAT.polynomial_set import mis
from FIAT.reference_element import UFCSimplex

HERE = os.path.dirname(os.path.abspath(__file__))


def stack(tab, sd, order):
    return np.stack([np.asarray(tab[a]) for k in range(order + 1) for a in mis(sd, k)])


def physical(sd, verts):
    ref = FIAT.ufc_simplex(sd)
    return UFCSimplex(ref.get_shape(), tuple(map(tuple, verts)), ref.get_topology())


def main():
    rng = np.random.default_rng(4004)
    out = {}
    cases = [("dg6tet", 3, lambda c: FIAT.DiscontinuousLagrange(c, 6), True), ("p6tri", 2, lambda c: FIAT.Lagrange(c, 6), True),
             ("p5tet", 3, lambda c: FIAT.Lagrange(c, 5), True), ("n4tri", 2, lambda c: FIAT.Nedelec(c, 4), False),
             ("on7int", 1, lambda c: polynomial_set.ONPolynomialSet(c, 7), False)]
    for name, sd, make, rebuild in cases:
        ref = np.array(FIAT.ufc_simplex(sd).get_vertices(), dtype=float)
        ncell, npts = 2, 5
        A = np.eye(sd) + 0.25 * rng.standard_normal((ncell, sd, sd))
        A[-1, :, 0] *= -1.0                                          # one negatively oriented cell
        verts = np.einsum("vd,red->rve", ref, A) + rng.standard_normal((ncell, 1, sd))
        e = rng.exponential(size=(ncell, npts, sd + 1))
        bary = e / e.sum(-1, keepdims=True)
        pts, ref_pts = np.einsum("rpv,rvd->rpd", bary, verts), np.einsum("rpv,vd->rpd", bary, ref)
        out[f"ho_{name}_verts"], out[f"ho_{name}_pts"], out[f"ho_{name}_refpts"] = verts, pts, ref_pts
        base = make(FIAT.ufc_simplex(sd))
        is_element = hasattr(base, "dual_basis")
        for order in (5, 6):
            for r in range(ncell):
                tab = base.tabulate(order, ref_pts[r]) if is_element else base.tabulate(ref_pts[r], order)
                out[f"ho_{name}_o{order}_ref{r}"] = stack(tab, sd, order)
                if rebuild:
                    out[f"ho_{name}_o{order}_phys{r}"] = stack(make(physical(sd, verts[r])).tabulate(order, pts[r]), sd, order)
    np.savez_compressed(os.path.join(HERE, "round4.npz"), **out)
    print(len(out), "arrays ->", os.path.join(HERE, "round4.npz"), os.path.getsize(os.path.join(HERE, "round4.npz")), "bytes")


if __name__ == "__main__":
    main()
