"""TEST INFRASTRUCTURE -- not part of the product.

The element definitions of fiat_amd/ (which dofs sit on which entity, which polynomial space) are host
bookkeeping around device arithmetic (expansion tabulation, Riesz assembly, Vandermonde solve).  To check that
bookkeeping in the ``-m "not gpu"`` suite, ``oracle_backend`` swaps the device entry points of
``fiat_amd.runtime`` for the CPU oracle (oracle/fiat_oracle.py) for the duration of one test.  Nothing outside
``tests/`` imports this module; the product has no CPU path."""
import math

import numpy as np
import pytest
import torch

from oracle import fiat_oracle as fo


class _Ctx:
    device = torch.device("cpu")
    handle = None
    num_cu, lds_per_cu, arch = 0, 0, "oracle"

    @staticmethod
    def get(device=None):
        return _Ctx


def _num_tables(sd, order):
    return math.comb(sd + order, sd)


class _SimplexPolySet:
    """runtime.SimplexPolySet with the oracle doing the arithmetic."""
    MAPPINGS = {"affine": 0, "covariant piola": 1, "contravariant piola": 2, "double covariant piola": 3,
                "double contravariant piola": 4, "covariant contravariant piola": 5}

    def __init__(self, sd, n, variant=None, scale=None, verts=None, coeffs=None, ndof=None, value_shape=(), ctx=None):
        self.sd, self.n, self.variant, self.scale = sd, n, variant, scale
        self.verts = np.asarray(fo.UFC_SIMPLEX[sd] if verts is None else verts, dtype=float).reshape(sd + 1, sd)
        self.nexp = math.comb(n + sd, sd)
        self.value_shape = tuple(value_shape)
        self.coeffs = None if coeffs is None else np.asarray(coeffs, dtype=float)
        self.ndof = self.nexp if coeffs is None else self.coeffs.shape[0]
        self.ctx = _Ctx

    def set_coeffs(self, coeffs):
        self.coeffs = np.asarray(coeffs, dtype=float)
        self.ndof = self.coeffs.shape[0]

    def tabulate_batch(self, order, pts, verts=None, out=None, stream=None, mapping=None):
        pts = np.asarray(pts, dtype=float)
        if verts is not None or mapping not in (None, "affine"):
            raise NotImplementedError("the CPU shim covers reference-cell tabulation only")
        alphas = fo.jet_indices(self.sd, order)
        reqs = []
        for P in pts:
            # the product keeps a given scale as it is (scale = get_scale(n) is applied by the facade)
            base = fo.expansion_tabulate(self.verts, self.n, P, order, self.scale, self.variant,
                                         single_cell=self.scale is None)
            if self.coeffs is None:
                tabs = [base[a] for a in alphas]
            else:
                C = self.coeffs.reshape(self.ndof, -1, self.nexp)
                tabs = [np.einsum("ick,kp->icp", C, base[a]).reshape((self.ndof,) + self.value_shape + (P.shape[0],))
                        for a in alphas]
            reqs.append(np.stack(tabs))
        return torch.as_tensor(np.stack(reqs))


class _LineLagrange:
    def __init__(self, nodes, ctx=None):
        self.nodes = np.asarray(nodes, dtype=float).reshape(-1)
        self.nn = len(self.nodes)
        self.ctx = _Ctx

    def tabulate_batch(self, order, pts, out=None, stream=None):
        pts = np.asarray(pts, dtype=float)
        res = []
        for P in pts:
            tab = fo.lagrange_line_tabulate(self.nodes, P.reshape(-1, 1), order)
            res.append(np.stack([tab[(r,)] for r in range(order + 1)]))
        return torch.as_tensor(np.stack(res))


def _riesz_assemble(wts, expvals, ctx=None):
    return torch.as_tensor(np.asarray(wts) @ np.asarray(expvals).T)


def _vandermonde_solve_batch(A, B, ctx=None, return_V=False):
    from fiat_amd import _lib
    A, B = np.asarray(A, dtype=float), np.asarray(B, dtype=float)
    if A.ndim == 2:
        A, B = A[None], B[None]
    X, V = np.empty_like(A), np.empty((A.shape[0], A.shape[1], A.shape[1]))
    for s in range(A.shape[0]):
        V[s] = A[s] @ B[s].T
        if np.linalg.cond(V[s]) > 1e14:
            raise _lib.LinAlgError("Singular Vandermonde matrix")
        X[s] = np.linalg.solve(V[s].T, B[s])
    X, V = torch.as_tensor(X), torch.as_tensor(V)
    return (X, V) if return_V else X


def _map_points(M, b, pts, ctx=None, stream=None):
    b = np.asarray(b, dtype=float).reshape(-1)
    M = np.asarray(M, dtype=float).reshape(len(b), -1)
    pts = np.asarray(pts, dtype=float)
    return torch.as_tensor(pts @ M.T + b if M.shape[1] else np.broadcast_to(b, pts.shape[:-1] + (len(b),)).copy())


def _tables_squared_norm(tables, weights, ctx=None, stream=None):
    t = np.asarray(tables, dtype=float)
    t = t.reshape(t.shape[0], t.shape[1], -1, t.shape[-1])
    return torch.as_tensor(np.einsum("nrcp,nrcp,p->nr", t, t, np.asarray(weights, dtype=float)))


class _MacroPolySet:
    """runtime.MacroPolySet with the oracle doing the arithmetic (macro_expansion_tabulate: binning, multiplicity, scatter)."""

    def __init__(self, sd, n, variant, scale, parent_verts, cell_verts, cell_node_map, nmacro, cell_scale=None,
                 coeffs=None, value_shape=(), ctx=None):
        self.sd, self.n, self.variant, self.scale = sd, n, variant, float(scale)
        self.parent = np.asarray(parent_verts, dtype=float).reshape(sd + 1, sd)
        self.cells = [np.asarray(c, dtype=float) for c in np.asarray(cell_verts, dtype=float)]
        self.cmap = np.asarray(cell_node_map)
        self.nmacro = int(nmacro)
        self.cell_scale = None if cell_scale is None else np.asarray(cell_scale, dtype=float)
        self.value_shape = tuple(value_shape)
        self.coeffs = None if coeffs is None else np.asarray(coeffs, dtype=float)
        self.ndof = self.nmacro if coeffs is None else self.coeffs.shape[0]

    def set_coeffs(self, coeffs):
        self.coeffs = np.asarray(coeffs, dtype=float)
        self.ndof = self.coeffs.shape[0]

    def out_shape(self, order, nreq, npts):
        return (nreq, _num_tables(self.sd, order), self.ndof) + self.value_shape + (npts,)

    def tabulate_batch(self, order, pts, verts=None, out=None, stream=None, mapping=None):
        if verts is not None or mapping not in (None, "affine"):
            raise NotImplementedError("test stand-in: macro elements on their own parent cell only")
        if self.cell_scale is not None and not np.allclose(self.cell_scale, 1.0):
            raise NotImplementedError("test stand-in: one scale for all sub-cells")
        pts = np.asarray(pts.cpu() if hasattr(pts, "cpu") else pts, dtype=float)
        res = np.zeros(self.out_shape(order, pts.shape[0], pts.shape[1]))
        for r in range(pts.shape[0]):
            base = fo.macro_expansion_tabulate(self.parent, self.cells, self.cmap, self.nmacro, self.n, pts[r], order, self.scale,
                                               self.variant)
            for t, a in enumerate(fo.jet_indices(self.sd, order)):
                res[r, t] = base[a] if self.coeffs is None else fo.contract(self.coeffs, {a: base[a]})[a]
        return torch.as_tensor(res)


@pytest.fixture
def oracle_backend(monkeypatch):
    from fiat_amd import runtime
    monkeypatch.setattr(runtime, "Context", _Ctx)
    monkeypatch.setattr(runtime, "SimplexPolySet", _SimplexPolySet)
    monkeypatch.setattr(runtime, "LineLagrange", _LineLagrange)
    monkeypatch.setattr(runtime, "MacroPolySet", _MacroPolySet)
    monkeypatch.setattr(runtime, "riesz_assemble", _riesz_assemble)
    monkeypatch.setattr(runtime, "vandermonde_solve_batch", _vandermonde_solve_batch)
    monkeypatch.setattr(runtime, "map_points", _map_points)
    monkeypatch.setattr(runtime, "tables_squared_norm", _tables_squared_norm)
    return runtime
