"""Gauss-Jacobi family quadratures on [-1,1] and the collapsed (Duffy) rule on
the (-1,1)^d simplex, as used by FIAT/quadrature.py:102,123,179."""
import numpy as np
from scipy.special import roots_jacobi


def gaussjacobi(n, a=0.0, b=0.0):
    x, w = roots_jacobi(n, a, b)
    return x, w


def lobattogaussjacobi(n, a=0.0, b=0.0):
    # endpoints +-1 and the n-2 roots of P_{n-2}^{(a+1,b+1)}; weights from
    # exactness on the Jacobi-weighted monomial moments.
    if n < 2:
        raise ValueError("Lobatto rule needs at least two points")
    xi = roots_jacobi(n - 2, a + 1, b + 1)[0] if n > 2 else np.zeros(0)
    x = np.concatenate([[-1.0], xi, [1.0]])
    m = max(n, 2)
    xg, wg = roots_jacobi(m + 1, a, b)
    V = np.vander(x, n, increasing=True).T
    rhs = np.array([np.dot(wg, xg ** k) for k in range(n)])
    w = np.linalg.solve(V, rhs)
    return x, w


def simplexgausslegendre(d, n):
    """Collapsed product rule on the (-1,1)^d simplex.  Cube coordinate j
    (j = 0..d-1) collapses coordinates i < j:
        x_i = (1 + e_i) * prod_{j>i} (1 - e_j)/2 - 1,
    so its Jacobian carries ((1-e_j)/2)^j -> Gauss-Jacobi(j,0) in e_j with
    weights / 2^j.  Weights sum to 2^d / d!."""
    import itertools
    rules = [roots_jacobi(n, j, 0) for j in range(d)]
    pts = []
    wts = []
    for idx in itertools.product(range(n), repeat=d):
        e = [rules[j][0][idx[j]] for j in range(d)]
        w = 1.0
        for j in range(d):
            w *= rules[j][1][idx[j]] / 2.0 ** j
        x = []
        for i in range(d):
            f = 1.0 + e[i]
            for j in range(i + 1, d):
                f *= (1.0 - e[j]) / 2.0
            x.append(f - 1.0)
        pts.append(x)
        wts.append(w)
    return np.array(pts), np.array(wts)
