"""1-D Jacobi polynomials, evaluated on the device (fx_jacobi_batch, csrc/jacobi_kernel.hpp).

Same call signatures and return shapes as FIAT/jacobi.py: ``eval_jacobi_batch(a, b, n, xs)`` (:47-74) returns the
(n + 1, npts) table of P_k^{(a,b)} at the points ``xs`` of shape (npts, 1); ``eval_jacobi_deriv_batch`` (:85-102) the
table of their derivatives of a given order; the scalar twins evaluate one polynomial.  ``jacobi_table`` is the
batched form that stays on the GPU."""
import numpy
import torch

from . import runtime


def jacobi_table(a, b, n, xs, order=0, out=None, stream=None):
    """Device tensor (n + 1, npts): row k = d^order/dx^order P_k^{(a,b)} at the points ``xs`` (any shape, flattened)."""
    ctx = runtime.Context.get()
    x = runtime._as_device(xs, ctx).reshape(-1)
    if out is None:
        out = torch.empty((n + 1, x.shape[0]), dtype=torch.float64, device=ctx.device)
    runtime.check(runtime.lib.fx_jacobi_batch(ctx.handle, float(a), float(b), int(n), int(order), x.shape[0],
                                              runtime._dev_ptr(x), runtime._dev_ptr(out), runtime._stream_ptr(stream)))
    return out


def _points(xs):
    xs = numpy.asarray(xs, dtype=float)
    return xs, (xs.shape[:-1] if xs.ndim > 1 else xs.shape)


def eval_jacobi_batch(a, b, n, xs):
    xs, shape = _points(xs)
    return runtime.fetch(jacobi_table(a, b, n, xs)).reshape((n + 1, *shape))


def eval_jacobi_deriv_batch(a, b, n, xs, order=1):
    xs = numpy.asarray(xs, dtype=float)
    return runtime.fetch(jacobi_table(a, b, n, xs, order=order)).reshape(n + 1, len(xs))


def eval_jacobi(a, b, n, x):
    """P_n^{(a,b)} at a point or an array of points (result has the shape of ``x``)."""
    x = numpy.asarray(x, dtype=float)
    row = runtime.fetch(jacobi_table(a, b, n, x))[n].reshape(x.shape)
    return float(row) if x.ndim == 0 else row


def eval_jacobi_deriv(a, b, n, x):
    x = numpy.asarray(x, dtype=float)
    row = runtime.fetch(jacobi_table(a, b, n, x, order=1))[n].reshape(x.shape)
    return float(row) if x.ndim == 0 else row
