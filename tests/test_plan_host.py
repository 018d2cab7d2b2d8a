"""Host logic of the HIP path, checked on CPU: the folded recurrence step table
and the C0 transform built by fiat_amd/csrc/plan.hpp (exported through the C
ABI) must reproduce the oracle when interpreted exactly as the device kernel
interprets them (fiat_amd/csrc/simplex_kernel.hpp: make_factors / apply_step)."""
import math

import numpy as np
import pytest

from fiat_amd import _lib
from oracle import fiat_oracle as fo


def interpret_plan(sd, n, order, X, J, variant, scale):
    """NumPy mirror of the device recurrence phase.  Returns (nexp, ntab, npts)."""
    phi0, ints, coefs = _lib.plan_steps(sd, n, variant, scale)
    npts = X.shape[1]
    nh = sd * (sd + 1) // 2
    ntab = [1, 1 + sd, 1 + sd + nh][order]
    nexp = math.comb(n + sd, sd)
    M = np.zeros((nexp, 1 + sd + nh, npts))
    M[0, 0] = phi0
    Xp = [X[i] for i in range(sd)] + [np.full(npts, -1.0)] * 2
    Jp = [np.outer(J[i], np.ones(npts)) for i in range(sd)] + [np.zeros((sd, npts))] * 2
    pairs = [(d1, d2) for d1 in range(sd) for d2 in range(d1, sd)]
    for (dst, cur, prv, codim), (A, B, C) in zip(ints, coefs):
        x, y, z = Xp[codim:codim + 3]
        dx, dy, dz = Jp[codim:codim + 3]
        fb = 0.5 * (y + z)
        fa = x + (fb + 1.0)
        fc = fb * fb
        dfb = 0.5 * (dy + dz)
        dfa = dx + dfb
        dfc = 2.0 * fb * dfb
        c = M[cur]
        p = M[prv] if prv >= 0 else np.zeros_like(c)
        f = A * fa - B * fb
        g = -C * fc
        df = A * dfa - B * dfb
        dg = -C * dfc
        new = np.zeros_like(c)
        new[0] = c[0] * f + p[0] * g
        for d in range(sd):
            new[1 + d] = c[1 + d] * f + c[0] * df[d] + p[1 + d] * g + p[0] * dg[d]
        for h, (d1, d2) in enumerate(pairs):
            t = c[1 + sd + h] * f + df[d1] * c[1 + d2] + df[d2] * c[1 + d1]
            t = t + p[1 + sd + h] * g + dg[d1] * p[1 + d2] + dg[d2] * p[1 + d1]
            t = t + (-C * 2.0 * dfb[d1] * dfb[d2]) * p[0]
            new[1 + sd + h] = t
        M[dst] = new
    M = M[:, :ntab]
    if variant == "bubble":
        T = _lib.plan_c0_transform(sd, n)
        M = np.einsum("ij,jtp->itp", T, M)
    return M


@pytest.mark.parametrize("sd", [1, 2, 3])
@pytest.mark.parametrize("variant", [None, "bubble", "dual"])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 6])
def test_plan_matches_oracle(sd, variant, n):
    if variant == "bubble" and n == 0:
        with pytest.raises(ValueError):
            _lib.plan_steps(sd, n, variant, 1.0)
        return
    rng = np.random.default_rng(sd * 100 + n)
    verts = fo.UFC_SIMPLEX[sd] + rng.uniform(-0.2, 0.2, size=(sd + 1, sd))
    e = rng.exponential(size=(9, sd + 1))
    bary = e / e.sum(axis=1, keepdims=True)
    pts = bary @ verts
    A, b = fo.affine_map(verts, fo.DEFAULT_SIMPLEX[sd])
    X = (pts @ A.T + b).T
    scale = 1.0 if variant == "bubble" else fo.expansion_scale(sd, n)
    order = 2
    got = interpret_plan(sd, n, order, X, A, variant, scale)
    if sd == 1 and variant is None:
        # the reference's 1-D default path is the Jacobi(k,k) shortcut; compare with
        # the generic recurrence restated by the oracle instead (same polynomials)
        tabs = fo.dubiner_tables(sd, n, order, X, A, scale, variant)
    else:
        tabs = fo.dubiner_tables(sd, n, order, X, A, scale, variant)
        if variant == "bubble":
            tabs = fo.c0_basis(sd, n, tabs)
    ref = np.concatenate(tabs, axis=1)
    assert got.shape == ref.shape
    err = np.max(np.abs(got - ref)) / max(1.0, np.max(np.abs(ref)))
    assert err < 1e-13, err


def test_one_d_default_matches_reference_shortcut():
    """sd=1 default variant: generic recurrence == LineExpansionSet shortcut
    (expansions.py:659-678) to rounding."""
    pts = np.linspace(0.05, 0.95, 7)[:, None]
    A, b = fo.affine_map(fo.UFC_SIMPLEX[1], fo.DEFAULT_SIMPLEX[1])
    X = (pts @ A.T + b).T
    got = interpret_plan(1, 5, 2, X, A, None, fo.expansion_scale(1, 5))
    ref = fo.expansion_tabulate(fo.UFC_SIMPLEX[1], 5, pts, 2)
    for k in range(3):
        assert np.allclose(got[:, k], ref[(k,)], rtol=1e-12, atol=1e-12)


def test_a_fragment_layout_doc():
    """The K padding of the MFMA operands is zero and T is a signed permutation
    plus corrections: row sums of T stay bounded, T is invertible."""
    for sd in (2, 3):
        T = _lib.plan_c0_transform(sd, 4)
        assert abs(np.linalg.det(T)) > 0.5


# ---- schedule of the cooperative (producer / consumer) kernel -------------------------
@pytest.mark.parametrize("sd,n", [(1, 1), (1, 6), (2, 1), (2, 3), (2, 6), (3, 1), (3, 2), (3, 3), (3, 6), (3, 8)])
@pytest.mark.parametrize("variant", [None, "bubble"])
def test_coop_schedule_invariants(sd, n, variant):
    """plan.hpp build_coop_plan: every member is published exactly once, in the K slot the
    coefficient fragments expect; K-step ranges are contiguous; every step finds its inputs in the
    producer's own chain state (interpreted here exactly as the device does)."""
    from fiat_amd import _lib
    C = _lib.plan_coop(sd, n, variant)
    phi0, ints, coefs = _lib.plan_steps(sd, n, variant)
    nexp = math.comb(n + sd, sd)
    producer = {int(d): (int(c), int(p), int(cd)) for d, c, p, cd in ints}     # member -> (cur, prv, codim)
    KS = C["KS"]
    kperm = C["kperm"]
    assert KS == (nexp + 3) // 4
    assert sorted(int(m) for m in kperm if m >= 0) == list(range(nexp))
    assert all(int(m) == -1 for m in kperm[nexp - 4 * KS:]) if nexp % 4 else True
    for w in range(4):
        E, ks = C["entries"][w], C["kstart"][w]
        assert ks[0] == 0 and ks[-1] == len(E) and np.all(np.diff(ks) >= 0)
        # chain state as in the kernel: per level the newest two members
        state = {0: [None, None], 1: [None, None], 2: [None, None]}
        for j in range(KS):
            for i in range(ks[j], ks[j + 1]):
                level, seed, publish, member = (int(v) for v in E[i])
                if level < 0:
                    assert publish >= 1
                    assert kperm[4 * j + publish - 1] == (member if seed == -1 else -1)
                    continue
                cur, prv, codim = producer[member]
                assert codim == level
                if seed != -2:                                  # chain start: newer = seed member, older = 0
                    seed_member = 0 if seed == -1 else state[seed][0]
                    assert prv == -1 and cur == seed_member, (w, i, member, cur, seed_member)
                    state[level] = [member, cur]
                else:
                    assert state[level][0] == cur and state[level][1] == prv, (w, i, member)
                    state[level] = [member, cur]
                if publish:
                    assert 1 <= publish <= 4 and kperm[4 * j + publish - 1] == member
