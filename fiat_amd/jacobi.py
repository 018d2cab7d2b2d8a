"""Jacobi polynomials on the host (FIAT/jacobi.py:47-102), used only while a dual set is being built: the
edge-moment weight functions of the Argyris element are Jacobi(2, 2) polynomials along the edge.  (The tabulation
hot path evaluates its 1-D recurrences on the device.)"""
import numpy


def eval_jacobi_batch(a, b, n, xs):
    """P_k^{(a,b)}(x), k = 0..n, by the three-term recurrence: (n + 1, npts) for xs of shape (npts, 1)."""
    xs = numpy.asarray(xs, dtype=float)
    x = xs.reshape(xs.shape[:-1])
    out = numpy.zeros((n + 1, *x.shape))
    out[0] = 1.0
    if n >= 1:
        out[1] = 0.5 * (a - b + (a + b + 2.0) * x)
    s = a + b
    for k in range(2, n + 1):
        den = 2.0 * k * (k + s) * (2.0 * k + s - 2.0)
        lin = (2.0 * k + s - 1.0) * (a * a - b * b) / den
        slope = (2.0 * k + s - 2.0) * (2.0 * k + s - 1.0) * (2.0 * k + s) / den
        back = 2.0 * (k + a - 1.0) * (k + b - 1.0) * (2.0 * k + s) / den
        out[k] = (lin + slope * x) * out[k - 1] - back * out[k - 2]
    return out


def eval_jacobi_deriv_batch(a, b, n, xs, order=1):
    """order-th derivatives of P_k^{(a,b)}, k = 0..n: d^m P_k^{(a,b)} = prod_{l<m} (k + a + b + 1 + l)/2 P_{k-m}^{(a+m,b+m)}."""
    xs = numpy.asarray(xs, dtype=float)
    out = numpy.zeros((n + 1, len(xs)))
    if n + 1 <= order:
        return out
    out[order:] = eval_jacobi_batch(a + order, b + order, n - order, xs)
    for k in range(order, n + 1):
        out[k] *= numpy.prod([0.5 * (a + b + k + 1 + l) for l in range(order)])
    return out
