"""Device runtime: contexts, element handles and the batched entry points.

PyTorch-ROCm is used only as the owner of device memory and streams; every
computation goes through the C ABI (include/fiat_amd.h) into the HIP kernels.
"""
import contextlib
import ctypes
import math
from ctypes import c_float, c_int, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import VARIANTS, check, host_ptr, lib

_contexts = {}


class Context:
    """One per GPU (fx_ctx).  ``Context.get()`` returns the context of the
    current torch device; creation fails loudly when no GPU is present."""

    def __init__(self, device_index):
        h = c_void_p()
        check(lib.fx_ctx_create(int(device_index), ctypes.byref(h)))
        self.handle = h
        self.device = torch.device("cuda", device_index)
        ncu, lds = c_int(0), c_int(0)
        name = ctypes.create_string_buffer(64)
        check(lib.fx_ctx_info(h, ctypes.byref(ncu), ctypes.byref(lds), name, 64))
        self.num_cu, self.lds_per_cu, self.arch = ncu.value, lds.value, name.value.decode()

    def check(self, stream=None):
        """Synchronise ``stream`` and raise FiatAmdError if a kernel of this context reported a scheduling failure."""
        check(lib.fx_ctx_check(self.handle, _stream_ptr(stream)))

    def set_policy(self, *names):
        """Kernel-selection policy (fx_ctx_set_policy): names from ``_lib.POLICY``; no names = the default."""
        flags = 0
        for n in names:
            flags |= _lib.POLICY[n]
        check(lib.fx_ctx_set_policy(self.handle, flags))

    def get_policy(self):
        flags = ctypes.c_uint(0)
        check(lib.fx_ctx_get_policy(self.handle, ctypes.byref(flags)))
        return {n for n, bit in _lib.POLICY.items() if flags.value & bit}

    @contextlib.contextmanager
    def policy(self, *names):
        """``with ctx.policy("no_fixed"): ...`` -- reach an A/B partner kernel through the same entry points."""
        before = self.get_policy()
        self.set_policy(*names)
        try:
            yield self
        finally:
            self.set_policy(*before)

    @staticmethod
    def get(device=None):
        if device is None:
            if not torch.cuda.is_available():
                # let the C library produce the error message
                idx = 0
            else:
                idx = torch.cuda.current_device()
        else:
            idx = torch.device(device).index or 0
        if idx not in _contexts:
            _contexts[idx] = Context(idx)
        return _contexts[idx]


def fetch(tensor, stream=None):
    """Device tensor -> NumPy array, then ``fx_ctx_check``: a dynamically scheduled kernel that gave up waiting for its
    work queue (csrc/work_queue.hpp) leaves incomplete tables -- the facade raises instead of returning them."""
    host = tensor.cpu().numpy()
    if tensor.is_cuda:
        Context.get(tensor.device).check(stream)
    return host


def _stream_ptr(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return c_void_p(stream.cuda_stream)


def _dev_ptr(t):
    return c_void_p(t.data_ptr())


def _as_device(x, ctx):
    if isinstance(x, torch.Tensor):
        if x.device != ctx.device or x.dtype != torch.float64 or not x.is_contiguous():
            x = x.to(device=ctx.device, dtype=torch.float64).contiguous()
        return x
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).to(ctx.device)


def num_tables(sd, order):
    return math.comb(sd + order, sd)


class SimplexPolySet:
    """Device-resident polynomial set over a simplex expansion set (fx_element):
    coeffs (ndof, *value_shape, nexp) over the Dubiner basis of degree n."""

    def __init__(self, sd, n, variant=None, scale=None, verts=None, coeffs=None, ndof=None,
                 value_shape=(), ctx=None):
        self.ctx = ctx or Context.get()
        self.sd, self.n, self.variant = sd, n, variant
        self.nexp = math.comb(n + sd, sd)
        self.value_shape = tuple(value_shape)
        self.vdim = int(np.prod(self.value_shape, dtype=int)) if self.value_shape else 1
        self.verts = None if verts is None else np.ascontiguousarray(verts, dtype=np.float64).reshape(sd + 1, sd)
        if coeffs is not None:
            coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
            ndof = coeffs.shape[0]
            assert coeffs.size == ndof * self.vdim * self.nexp
        elif ndof is None:
            ndof = self.nexp
        self.ndof = ndof
        h = c_void_p()
        check(lib.fx_element_create(self.ctx.handle, sd, n, VARIANTS[variant],
                                    -1.0 if scale is None else float(scale),
                                    None if self.verts is None else host_ptr(self.verts),
                                    ndof, self.vdim, None if coeffs is None else host_ptr(coeffs),
                                    ctypes.byref(h)))
        self.handle = h

    def set_coeffs(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        ndof = coeffs.shape[0]
        assert coeffs.size == ndof * self.vdim * self.nexp
        check(lib.fx_element_set_coeffs(self.handle, ndof, self.vdim, host_ptr(coeffs)))
        self.ndof = ndof

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.fx_element_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def out_shape(self, order, nreq, npts):
        return (nreq, num_tables(self.sd, order), self.ndof) + self.value_shape + (npts,)

    MAPPINGS = {"affine": 0, "covariant piola": 1, "contravariant piola": 2, "double covariant piola": 3,
                "double contravariant piola": 4, "covariant contravariant piola": 5}

    def tabulate_batch(self, order, pts, verts=None, out=None, stream=None, mapping=None):
        """pts (nreq, npts, sd) -> (nreq, ntab, ndof, *value_shape, npts) on the GPU.

        With per-request cells ``verts`` the derivatives are taken with respect to the physical
        coordinates; ``mapping`` ("affine", "covariant piola", "contravariant piola": the names
        of FiniteElement.mapping()) additionally pushes vector-valued functions forward to the
        physical cells (fx_pushforward_batch)."""
        ctx = self.ctx
        pts = _as_device(pts, ctx)
        if pts.dim() != 3 or pts.shape[2] != self.sd:
            raise ValueError(f"points must have shape (nreq, npts, {self.sd}), got {tuple(pts.shape)}")
        nreq, npts = pts.shape[0], pts.shape[1]
        if verts is not None:
            verts = _as_device(verts, ctx)
            if tuple(verts.shape) != (nreq, self.sd + 1, self.sd):
                raise ValueError("verts must have shape (nreq, sd+1, sd)")
        shape = self.out_shape(order, nreq, npts)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=ctx.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out has the wrong shape/dtype/layout")
        if mapping is not None and mapping not in self.MAPPINGS:
            raise ValueError(f"unknown mapping {mapping!r}")
        if mapping not in (None, "affine") and verts is None:
            raise ValueError("a Piola push-forward needs the physical cells (verts)")
        if mapping in (None, "affine"):
            check(lib.fx_tabulate_batch(ctx.handle, self.handle, int(order), nreq, npts, _dev_ptr(pts),
                                        None if verts is None else _dev_ptr(verts), _dev_ptr(out), _stream_ptr(stream)))
        else:   # fused into the kernel's output stage where the shape allows, else a second pass
            check(lib.fx_tabulate_batch_mapped(ctx.handle, self.handle, self.MAPPINGS[mapping], int(order), nreq, npts,
                                               _dev_ptr(pts), _dev_ptr(verts), _dev_ptr(out), _stream_ptr(stream)))
        return out

    def pushforward_batch(self, order, out, verts, mapping):
        """In-place push-forward of tables produced by ``tabulate_batch(order, pts, verts=verts)``
        (fx_pushforward_batch: the separate pass; ``tabulate_batch(..., mapping=...)`` fuses it where it can)."""
        if mapping not in self.MAPPINGS:
            raise ValueError(f"unknown mapping {mapping!r}")
        verts = _as_device(verts, self.ctx)
        nreq, npts = out.shape[0], out.shape[-1]
        check(lib.fx_pushforward_batch(self.ctx.handle, self.handle, self.MAPPINGS[mapping], int(order), nreq, npts,
                                       _dev_ptr(verts), _dev_ptr(out), _stream_ptr(None)))
        return out

    def tabulate_batch_shared(self, order, ref_pts, verts, mapping="affine", out=None, stream=None):
        """One point set ``ref_pts`` (npts, sd) on the element's own cell, pushed forward to the
        cells ``verts`` (nreq, sd+1, sd): (nreq, ntab, ndof, *value_shape, npts) on the GPU, equal to
        ``tabulate_batch(order, F_r(ref_pts), verts, mapping=mapping)`` (fx_tabulate_batch_shared)."""
        ctx = self.ctx
        ref_pts = _as_device(ref_pts, ctx)
        verts = _as_device(verts, ctx)
        if ref_pts.dim() != 2 or ref_pts.shape[1] != self.sd:
            raise ValueError(f"reference points must have shape (npts, {self.sd})")
        if verts.dim() != 3 or tuple(verts.shape[1:]) != (self.sd + 1, self.sd):
            raise ValueError("verts must have shape (nreq, sd+1, sd)")
        if mapping not in self.MAPPINGS:
            raise ValueError(f"unknown mapping {mapping!r}")
        nreq, npts = verts.shape[0], ref_pts.shape[0]
        shape = self.out_shape(order, nreq, npts)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=ctx.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out has the wrong shape/dtype/layout")
        check(lib.fx_tabulate_batch_shared(ctx.handle, self.handle, self.MAPPINGS[mapping], int(order), nreq, npts,
                                           _dev_ptr(ref_pts), _dev_ptr(verts), _dev_ptr(out), _stream_ptr(stream)))
        return out

    def kernel_name(self, order, nreq, npts, has_verts=False, instance=False, mapping=None):
        """Device kernel ``tabulate_batch`` launches for this request shape (fx_plan_kernel); ``instance``: with the
        registry instance of the stacked-matrix kernel; ``mapping``: the Piola map asked for with the tabulation."""
        buf = ctypes.create_string_buffer(96)
        code = self.MAPPINGS[mapping] if mapping else 0
        flags = int(bool(has_verts)) | (2 if instance else 0) | ((code << 2) if code in (1, 2) else 0)
        check(lib.fx_plan_kernel(self.ctx.handle, self.handle, int(order), int(nreq), int(npts), flags, buf, 96))
        return buf.value.decode()

    def time_tabulate_batch(self, order, pts, verts, out, reps, stream=None):
        """Average kernel milliseconds over ``reps`` launches (HIP events on the launch stream)."""
        ms = c_float(0.0)
        check(lib.fx_time_tabulate_batch(self.ctx.handle, self.handle, int(order), pts.shape[0], pts.shape[1],
                                         _dev_ptr(pts), None if verts is None else _dev_ptr(verts), _dev_ptr(out),
                                         _stream_ptr(stream), int(reps), ctypes.byref(ms)))
        return ms.value


class MacroPolySet:
    """Device-resident polynomial set over the expansion set of a macro cell (fx_macro_element):
    coeffs (ndof, *value_shape, nmacro) over the members of the complex; the kernel bins every point to
    its sub-cell(s), runs the sub-cell's recurrence and contracts (FIAT/expansions.py:449-490)."""

    def __init__(self, sd, n, variant, scale, parent_verts, cell_verts, cell_node_map, nmacro, cell_scale=None,
                 coeffs=None, value_shape=(), ctx=None):
        self.ctx = ctx or Context.get()
        self.sd, self.n, self.variant = sd, n, variant
        self.nexp = math.comb(n + sd, sd)
        self.nmacro = int(nmacro)
        self.value_shape = tuple(value_shape)
        self.vdim = int(np.prod(self.value_shape, dtype=int)) if self.value_shape else 1
        parent = np.ascontiguousarray(parent_verts, dtype=np.float64).reshape(sd + 1, sd)
        cells = np.ascontiguousarray(cell_verts, dtype=np.float64)
        ncell = cells.shape[0]
        if cells.shape != (ncell, sd + 1, sd):
            raise ValueError("cell_verts must have shape (ncell, sd+1, sd)")
        cmap = np.ascontiguousarray(cell_node_map, dtype=np.int32)
        if cmap.shape != (ncell, self.nexp):
            raise ValueError(f"cell_node_map must have shape ({ncell}, {self.nexp}), got {cmap.shape}")
        cs = None if cell_scale is None else np.ascontiguousarray(cell_scale, dtype=np.float64).reshape(ncell)
        if coeffs is not None:
            coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
            self.ndof = coeffs.shape[0]
            assert coeffs.size == self.ndof * self.vdim * self.nmacro
        else:
            self.ndof = self.nmacro
        self.ncell = ncell
        h = c_void_p()
        check(lib.fx_macro_element_create(self.ctx.handle, sd, n, VARIANTS[variant], float(scale), host_ptr(parent), ncell,
                                          host_ptr(cells), self.nmacro, host_ptr(cmap),
                                          None if cs is None else host_ptr(cs), self.ndof, self.vdim,
                                          None if coeffs is None else host_ptr(coeffs), ctypes.byref(h)))
        self.handle = h

    def set_coeffs(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        ndof = coeffs.shape[0]
        assert coeffs.size == ndof * self.vdim * self.nmacro
        check(lib.fx_macro_element_set_coeffs(self.handle, ndof, self.vdim, host_ptr(coeffs)))
        self.ndof = ndof

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.fx_macro_element_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def out_shape(self, order, nreq, npts):
        return (nreq, num_tables(self.sd, order), self.ndof) + self.value_shape + (npts,)

    def tabulate_batch(self, order, pts, verts=None, out=None, stream=None, mapping=None):
        """pts (nreq, npts, sd) -> (nreq, ntab, ndof, *value_shape, npts) on the GPU; ``verts``: per-request
        parent cells (points are binned after pulling them back to the element's parent cell)."""
        if mapping not in (None, "affine"):
            raise NotImplementedError("Piola push-forwards of macro elements")
        ctx = self.ctx
        pts = _as_device(pts, ctx)
        if pts.dim() != 3 or pts.shape[2] != self.sd:
            raise ValueError(f"points must have shape (nreq, npts, {self.sd}), got {tuple(pts.shape)}")
        nreq, npts = pts.shape[0], pts.shape[1]
        if verts is not None:
            verts = _as_device(verts, ctx)
            if tuple(verts.shape) != (nreq, self.sd + 1, self.sd):
                raise ValueError("verts must have shape (nreq, sd+1, sd)")
        shape = self.out_shape(order, nreq, npts)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=ctx.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out has the wrong shape/dtype/layout")
        check(lib.fx_macro_tabulate_batch(ctx.handle, self.handle, int(order), nreq, npts, _dev_ptr(pts),
                                          None if verts is None else _dev_ptr(verts), _dev_ptr(out), _stream_ptr(stream)))
        return out

    _SMALL = {(2, 1): 2, (2, 2): 2, (2, 3): 2, (3, 1): 2, (3, 2): 2, (3, 3): 1}   # (sd, n): highest order (api.hip)

    def kernel_name(self, order, nreq, npts, has_verts=False):
        """Kernel a request shape maps to (mirrors the selection in fx_macro_tabulate_batch)."""
        small = (order <= self._SMALL.get((self.sd, self.n), -1) and npts <= 64
                 and "no_macro_small" not in self.ctx.get_policy())
        return "fxk::tabulate_macro_small" if small else "fxk::tabulate_simplex_kernel<MACRO>"


def collapsed_quadrature(sd, m, verts=None, ctx=None, stream=None):
    """Collapsed Gauss-Jacobi rule with ``m`` points per direction on the simplex ``verts``
    ((sd+1, sd) array, default: the UFC simplex), produced on the device:
    (points (m**sd, sd), weights (m**sd,)) as CUDA tensors (fx_collapsed_quadrature)."""
    ctx = ctx or Context.get()
    n = int(m) ** int(sd)
    pts = torch.empty((n, sd), dtype=torch.float64, device=ctx.device)
    wts = torch.empty((n,), dtype=torch.float64, device=ctx.device)
    hv = None
    if verts is not None:
        hv = np.ascontiguousarray(verts, dtype=np.float64)
        if hv.shape != (sd + 1, sd):
            raise ValueError("verts must have shape (sd+1, sd)")
    check(lib.fx_collapsed_quadrature(ctx.handle, int(sd), int(m), None if hv is None else host_ptr(hv), _dev_ptr(pts),
                                      _dev_ptr(wts), _stream_ptr(stream)))
    return pts, wts


def classify_tables(tables, rtol=1e-5, ctx=None, stream=None):
    """Per table of a device tensor ``tables[..., rows, npts]``: (max |x|, max(|x - x[..., :1]| - rtol |x[..., :1]|))
    as a device tensor ``[..., 2]`` (fx_classify_tables; what finat/fiat_elements.py:92-111 asserts)."""
    ctx = ctx or Context.get()
    tables = _as_device(tables, ctx)
    if tables.dim() < 2:
        raise ValueError("tables must have shape (..., rows, npts)")
    rows, npts = tables.shape[-2:]
    lead = tuple(tables.shape[:-2])
    ntables = int(np.prod(lead, dtype=np.int64)) if lead else 1
    stats = torch.empty(lead + (2,), dtype=torch.float64, device=ctx.device)
    check(lib.fx_classify_tables(ctx.handle, ntables, int(rows), int(npts), float(rtol), _dev_ptr(tables), _dev_ptr(stats),
                                 _stream_ptr(stream)))
    return stats


def tables_squared_norm(tables, weights, ctx=None, stream=None):
    """``tables[n, rows, *value_shape, npts]`` (order-0 tables at the points of one rule) and the rule's ``weights[npts]``
    -> device tensor ``[n, rows]`` of squared L2 norms (fx_tables_squared_norm; FIAT/finite_element.py:250-260)."""
    ctx = ctx or Context.get()
    tables = _as_device(tables, ctx)
    weights = _as_device(weights, ctx)
    if tables.dim() < 3 or weights.dim() != 1 or weights.shape[0] != tables.shape[-1]:
        raise ValueError("tables must have shape (n, rows, ..., npts) and weights (npts,)")
    n, rows, npts = int(tables.shape[0]), int(tables.shape[1]), int(tables.shape[-1])
    vdim = int(np.prod(tables.shape[2:-1], dtype=np.int64)) if tables.dim() > 3 else 1
    out = torch.empty((n, rows), dtype=torch.float64, device=ctx.device)
    check(lib.fx_tables_squared_norm(ctx.handle, n, rows, vdim, npts, _dev_ptr(tables), _dev_ptr(weights), _dev_ptr(out),
                                     _stream_ptr(stream)))
    return out


def tables_point_major(tables, out=None, ctx=None, stream=None):
    """``tables[..., rows, npts]`` -> a new contiguous device tensor ``[..., npts, rows]`` (fx_tables_point_major)."""
    ctx = ctx or Context.get()
    tables = _as_device(tables, ctx)
    if tables.dim() < 2:
        raise ValueError("tables must have shape (..., rows, npts)")
    rows, npts = tables.shape[-2:]
    lead = tuple(tables.shape[:-2])
    ntables = int(np.prod(lead, dtype=np.int64)) if lead else 1
    if out is None:
        out = torch.empty(lead + (npts, rows), dtype=torch.float64, device=ctx.device)
    check(lib.fx_tables_point_major(ctx.handle, ntables, int(rows), int(npts), _dev_ptr(tables), _dev_ptr(out),
                                    _stream_ptr(stream)))
    return out


def map_points(M, b, pts, ctx=None, stream=None):
    """Affine image of a batch of points on the device: pts (..., din) -> (..., dout) = pts M^T + b
    (fx_map_points; the entity transform of ``tabulate(..., entity=)``)."""
    ctx = ctx or Context.get()
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    dout = b.shape[0]
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(dout, -1)
    din = M.shape[1]
    if isinstance(pts, torch.Tensor):
        lead = tuple(pts.shape[:-1])
        if pts.shape[-1] != din:
            raise ValueError(f"points must have {din} coordinates, got {pts.shape[-1]}")
        src = _as_device(pts, ctx) if din else None
    else:
        pts = np.asarray(pts, dtype=np.float64)
        lead = tuple(pts.shape[:-1])
        if pts.shape[-1] != din:
            raise ValueError(f"points must have {din} coordinates, got {pts.shape[-1]}")
        src = _as_device(pts, ctx) if din else None
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    out = torch.empty(lead + (dout,), dtype=torch.float64, device=ctx.device)
    check(lib.fx_map_points(ctx.handle, din, dout, host_ptr(M) if din else None, host_ptr(b), n,
                            None if src is None else _dev_ptr(src), _dev_ptr(out), _stream_ptr(stream)))
    return out


def table_outer(order, sdA, sdB, tabA, tabB, out=None, ctx=None, stream=None):
    """Tensor product of two tabulated factors (fx_table_outer_batch): tabA (nreq, ntabA, ndofA, [vdimA,] npts),
    tabB (nreq, ntabB, ndofB, [vdimB,] npts) -> (nreq, ntab, ndofA * ndofB, [vdim,] npts), the tables of all
    |alpha| <= order of the product in mis() order."""
    ctx = ctx or Context.get()
    tabA, tabB = _as_device(tabA, ctx), _as_device(tabB, ctx)
    nreq, npts = tabA.shape[0], tabA.shape[-1]
    if tabB.shape[0] != nreq or tabB.shape[-1] != npts:
        raise ValueError("factor tables must share the request and point axes")
    vdimA = tabA.shape[3] if tabA.dim() == 5 else 1
    vdimB = tabB.shape[3] if tabB.dim() == 5 else 1
    if vdimA > 1 and vdimB > 1:
        raise NotImplementedError("tabulate does not support two vector-valued inputs")
    rowsA, rowsB = tabA.shape[2], tabB.shape[2]
    if tabA.shape[1] != num_tables(sdA, order) or tabB.shape[1] != num_tables(sdB, order):
        raise ValueError("factor tables must hold all derivative tables up to the product's order")
    vshape = (max(vdimA, vdimB),) if max(tabA.dim(), tabB.dim()) == 5 else ()
    shape = (nreq, num_tables(sdA + sdB, order), rowsA * rowsB) + vshape + (npts,)
    if out is None:
        out = torch.empty(shape, dtype=torch.float64, device=ctx.device)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError("out has the wrong shape/layout")
    check(lib.fx_table_outer_batch(ctx.handle, int(order), int(sdA), int(sdB), nreq, npts, rowsA, vdimA, rowsB, vdimB,
                                   _dev_ptr(tabA), _dev_ptr(tabB), _dev_ptr(out), _stream_ptr(stream)))
    return out


def riesz_assemble(wts, expvals, ctx=None):
    """mat[i, k] = sum_q wts[i, q] expvals[k, q] on the device."""
    ctx = ctx or Context.get()
    wts = _as_device(wts, ctx)
    expvals = _as_device(expvals, ctx)
    nrows, nq = wts.shape
    nexp = expvals.shape[0]
    assert expvals.shape[1] == nq
    mat = torch.empty((nrows, nexp), dtype=torch.float64, device=ctx.device)
    check(lib.fx_riesz_assemble(ctx.handle, nrows, nq, nexp, _dev_ptr(wts), _dev_ptr(expvals), _dev_ptr(mat),
                                _stream_ptr(None)))
    return mat


def vandermonde_solve_batch(A, B, ctx=None, return_V=False):
    """X = solve((A B^T)^T, B) per system; raises LinAlgError on a zero pivot
    ("Singular Vandermonde matrix", finite_element.py:156)."""
    ctx = ctx or Context.get()
    A = _as_device(A, ctx)
    B = _as_device(B, ctx)
    if A.dim() == 2:
        A, B = A[None], B[None]
    nsys, ndof, m = A.shape
    assert B.shape == A.shape
    X = torch.empty_like(A)
    V = torch.empty((nsys, ndof, ndof), dtype=torch.float64, device=ctx.device) if return_V else None
    info = torch.zeros((nsys,), dtype=torch.int32, device=ctx.device)
    check(lib.fx_vandermonde_solve_batch(ctx.handle, nsys, ndof, m, _dev_ptr(A), _dev_ptr(B), _dev_ptr(X),
                                         None if V is None else _dev_ptr(V), _dev_ptr(info), _stream_ptr(None)))
    if int(info.abs().max().item()) != 0:
        raise _lib.LinAlgError("Singular Vandermonde matrix")
    return (X, V) if return_V else X


class LineLagrange:
    """1-D Lagrange basis on given nodes (fx_line_element)."""

    def __init__(self, nodes, ctx=None):
        self.ctx = ctx or Context.get()
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float64).reshape(-1)
        h = c_void_p()
        check(lib.fx_line_element_create(self.ctx.handle, len(self.nodes), host_ptr(self.nodes), ctypes.byref(h)))
        self.handle = h
        self.nn = len(self.nodes)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.fx_line_element_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def tabulate_batch(self, order, pts, out=None, stream=None):
        """pts (nreq, npts) -> (nreq, order+1, nn, npts)."""
        ctx = self.ctx
        pts = _as_device(pts, ctx)
        nreq, npts = pts.shape
        if out is None:
            out = torch.empty((nreq, order + 1, self.nn, npts), dtype=torch.float64, device=ctx.device)
        check(lib.fx_line_tabulate_batch(ctx.handle, self.handle, int(order), nreq, npts, _dev_ptr(pts), _dev_ptr(out),
                                         _stream_ptr(stream)))
        return out


def prism_tabulate_batch(tri, line, order, pts, out=None, stream=None):
    """(element on a triangle) x (1-D Lagrange element) on the fused prism kernel (fx_prism_tabulate_batch): ``tri`` a
    SimplexPolySet of spatial dimension 2, ``line`` a LineLagrange, pts (nreq, npts, 3) ->
    (nreq, ntab, ndofA * ndofB, [vdim,] npts); None when the shape has no instance (the caller takes the general route)."""
    ctx = tri.ctx
    pts = _as_device(pts, ctx)
    if pts.dim() != 3 or pts.shape[2] != 3:
        raise ValueError(f"points must have shape (nreq, npts, 3), got {tuple(pts.shape)}")
    nreq, npts = int(pts.shape[0]), int(pts.shape[1])
    shapeA = tri.out_shape(order, nreq, npts)                      # (nreq, ntabA, ndofA, [vdim,] npts)
    shape = (nreq, num_tables(3, order), shapeA[2] * line.nn) + tuple(shapeA[3:])
    if out is None:
        out = torch.empty(shape, dtype=torch.float64, device=ctx.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous():
        raise ValueError("out has the wrong shape/dtype/layout")
    rc = lib.fx_prism_tabulate_batch(ctx.handle, tri.handle, line.handle, int(order), nreq, npts, _dev_ptr(pts), _dev_ptr(out),
                                     _stream_ptr(stream))
    if rc == _lib.FX_ENOTIMPL:
        return None
    check(rc)
    return out


def tensor_tabulate_batch(factors, order, pts, out=None, stream=None, grid=False):
    """Tensor-product tabulation of 1..3 LineLagrange factors.
    grid=False: pts (nreq, npts, nf); grid=True: pts (nreq, nf, q) 1-D coordinates."""
    ctx = factors[0].ctx
    nf = len(factors)
    pts = _as_device(pts, ctx)
    arr = (c_void_p * nf)(*[f.handle for f in factors])
    nbf = int(np.prod([f.nn for f in factors]))
    ntab = num_tables(nf, order)
    if grid:
        nreq, nf_, q = pts.shape
        assert nf_ == nf
        npts = q ** nf
    else:
        nreq, npts, nf_ = pts.shape
        assert nf_ == nf
    if out is None:
        out = torch.empty((nreq, ntab, nbf, npts), dtype=torch.float64, device=ctx.device)
    if grid:
        check(lib.fx_tensor_tabulate_grid_batch(ctx.handle, nf, arr, int(order), nreq, q, _dev_ptr(pts), _dev_ptr(out),
                                                _stream_ptr(stream)))
    else:
        check(lib.fx_tensor_tabulate_batch(ctx.handle, nf, arr, int(order), nreq, npts, _dev_ptr(pts), _dev_ptr(out),
                                           _stream_ptr(stream)))
    return out
