"""Recursive simplex nodes (Isaac 2020, eq. (3.1)): barycentric coordinates of
lattice multi-index ``alpha`` (|alpha| = n) in d dimensions are the weighted
average of the (d-1)-dimensional rule applied on each facet, weights taken from
the 1-D node family.  For the equispaced family this is exactly alpha / n."""
import numpy as np
from scipy.special import roots_jacobi


class _Family:
    def __init__(self, fn):
        self._fn = fn
        self._cache = {}

    def __getitem__(self, n):
        if n not in self._cache:
            self._cache[n] = self._fn(n)
        return self._cache[n]


def _equi(n):
    return np.array([0.5]) if n == 0 else np.linspace(0.0, 1.0, n + 1)


def _equi_interior(n):
    return (np.arange(n + 1) + 0.5) / (n + 1)


def _lgl(n):
    if n == 0:
        return np.array([0.5])
    if n == 1:
        return np.array([0.0, 1.0])
    x, _ = roots_jacobi(n - 1, 1, 1)
    return np.concatenate([[0.0], (x + 1) / 2, [1.0]])


def _gl(n):
    return (roots_jacobi(n + 1, 0, 0)[0] + 1) / 2


def _lgc(n):
    if n == 0:
        return np.array([0.5])
    return (1 - np.cos(np.pi * np.arange(n + 1) / n)) / 2


def _gc(n):
    return (1 - np.cos(np.pi * (2 * np.arange(n + 1) + 1) / (2 * n + 2))) / 2


_families = {"equi": _equi, "equi_interior": _equi_interior, "lgl": _lgl,
             "gl": _gl, "lgc": _lgc, "gc": _gc}


def _decode_family(family):
    if isinstance(family, _Family):
        return family
    return _Family(_families[family])


def _recursive(d, n, alpha, family):
    xn = family[n]
    b = np.zeros(d + 1)
    if d == 1:
        b[:] = xn[[alpha[0], alpha[1]]]
        return b
    weight = 0.0
    for i in range(d + 1):
        w = xn[n - alpha[i]]
        br = _recursive(d - 1, n - alpha[i], alpha[:i] + alpha[i + 1:], family)
        b[:i] += w * br[:i]
        b[i + 1:] += w * br[i:]
        weight += w
    return b / weight
