"""Pin the C restatement (oracle/fiat_oracle.c, the bench's CPU baseline and the
full-batch checker) against the NumPy oracle and the reference's golden tables."""
import numpy as np
import pytest

from oracle import c_oracle, fiat_oracle as fo


def stacked(tab, sd, order):
    return np.stack([tab[a] for a in fo.jet_indices(sd, order)])


def relerr(x, ref):
    assert x.shape == ref.shape, (x.shape, ref.shape)
    return float(np.max(np.abs(x - ref)) / max(1.0, np.max(np.abs(ref))))


@pytest.mark.parametrize("sd", [1, 2, 3])
@pytest.mark.parametrize("variant", [None, "bubble", "dual"])
def test_expansion_tables(golden, sd, variant):
    g = golden("expansion")
    checked = 0
    for ci in (0, 1):
        verts, pts = g[f"verts_sd{sd}_c{ci}"], g[f"cpts_sd{sd}_c{ci}"]
        for n in (0, 1, 2, 3, 4, 6):
            for order in (0, 1, 2):
                key = f"exp_sd{sd}_c{ci}_{variant}_n{n}_o{order}"
                if key not in g:
                    continue
                nexp = g[key].shape[1]
                got = c_oracle.tabulate_batch(verts, n, np.eye(nexp), order, pts[None], variant=variant)[0]
                assert relerr(got, g[key]) < 1e-12, key
                checked += 1
    assert checked >= 20


def test_c2_and_physical_cells(golden):
    g = golden("elements")
    co = g["c2_p3tet_q6_coeffs"]
    got = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, g["c2_p3tet_rand_pts"], scale=1, variant="bubble")
    assert relerr(got, g["c2_p3tet_rand_tab"]) < 1e-12
    got = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 1, g["c2_phys_pts"], verts=g["c2_phys_verts"], scale=1,
                                  variant="bubble")
    assert relerr(got, g["c2_phys_tab"]) < 1e-11
    got = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 3, co, 2, g["c2_p3tet_rand_pts"][:1], scale=1, variant="bubble")
    assert relerr(got[0], g["c2_p3tet_o2_tab"]) < 1e-11


def test_vector_valued_and_dg6(golden):
    g = golden("elements")
    for name in ("n2", "rt2"):
        co = g[f"c3_{name}tet_q6_coeffs"]
        got = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 2, co, 1, g["tet_q6_pts"][None])[0]
        ref = g[f"c3_{name}tet_q6_tab"]
        assert relerr(got.reshape(ref.shape), ref) < 1e-12
    co = g["c4_dg6tet_q6_coeffs"]
    got = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[3], 6, co, 2, g["tet_q6_pts"][None])[0]
    assert relerr(got, g["c4_dg6tet_q6_tab"]) < 1e-10


def test_matches_numpy_oracle_on_a_batch():
    rng = np.random.default_rng(11)
    co, _, _ = fo.lagrange_coeffs(fo.UFC_SIMPLEX[2], 3)
    e = rng.exponential(size=(50, 7, 3))
    pts = (e / e.sum(-1, keepdims=True))[..., 1:].copy()
    got = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[2], 3, co, 2, pts, scale=1, variant="bubble", nthreads=2)
    for r in (0, 17, 49):
        ref = stacked(fo.element_tabulate(fo.UFC_SIMPLEX[2], 3, co, 2, pts[r], 1, "bubble"), 2, 2)
        assert relerr(got[r], ref) < 1e-12
    assert c_oracle.max_threads() >= 1
