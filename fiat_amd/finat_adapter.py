"""The FInAT side of the tabulation boundary, served from GPU-resident tables.

FInAT is the caller of ``element.tabulate`` (finat/fiat_elements.py:69): it checks each FIAT table
(derivative order == degree: constant over the points; > degree: zero; :92-111), reshapes it to
``index_shape + value_shape + point_shape`` and wraps it in a GEM literal.  GEM is out of scope
here; this module produces exactly the arrays (and the facts about them) that would become those
literals, for one point set through the FIAT-style ``tabulate`` or for a whole batch of requests
without leaving the device, plus the run-time-tabulated hook (finat/runtime_tabulated.py:68-95),
whose tables are kernel arguments in point-major layout, and the per-factor tables a
sum-factorised tensor-product evaluation multiplies (finat/tensor_product.py:120-144).

Parity note: ``finat`` cannot be imported in the build container (it needs ``ufl``, which is not
installed), so this row is pinned by the reference's source text only — "parity unpinned" — and by
the tables themselves, which are the parity-checked FIAT tabulations.
"""
import numpy

from . import runtime
from .polynomial_set import mis

# numpy.allclose defaults, which finat/fiat_elements.py:101,106 relies on
_RTOL, _ATOL = 1.0e-5, 1.0e-8

POINTWISE, CELLWISE_CONSTANT, ZERO = "pointwise", "cellwise_constant", "zero"


class PointSet:
    """A vector of N points of shape (N, D) (finat/point_set.py:138-149)."""

    def __init__(self, points):
        points = numpy.asarray(points)
        assert len(points.shape) == 2
        self.points = points

    @property
    def dimension(self):
        return self.points.shape[-1]

    @property
    def extents(self):
        """Extents of the point indices (finat/point_set.py:158-160)."""
        return tuple(self.points.shape[:-1])


class Table:
    """What FInAT would hand to ``gem.as_gem`` for one multi-index: ``array`` of shape
    ``index_shape + value_shape + point_shape`` (leading ``nreq`` axis in batch mode; NumPy array or
    device tensor) and ``kind``: "pointwise" (point_shape = the point extents), "cellwise_constant"
    or "zero" (point_shape = ())."""

    def __init__(self, array, kind):
        self.array = array
        self.kind = kind

    @property
    def shape(self):
        return tuple(self.array.shape)

    def __repr__(self):
        return f"Table({self.kind}, shape={self.shape})"


class FiatElement:
    """finat/fiat_elements.py:13-123 without GEM: the element metadata FInAT forwards and
    ``basis_evaluation`` returning :class:`Table` objects instead of GEM expressions."""

    def __init__(self, fiat_element):
        self._element = fiat_element

    @property
    def cell(self):
        return self._element.get_reference_element()

    @property
    def complex(self):
        return self._element.get_reference_complex()

    @property
    def degree(self):
        return self._element.degree()

    @property
    def formdegree(self):
        return self._element.get_formdegree()

    def entity_dofs(self):
        return self._element.entity_dofs()

    def entity_closure_dofs(self):
        return self._element.entity_closure_dofs()

    def space_dimension(self):
        return self._element.space_dimension()

    @property
    def index_shape(self):
        return (self.space_dimension(),)

    @property
    def value_shape(self):
        return self._element.value_shape()

    @property
    def mapping(self):
        mappings = set(self._element.mapping())
        if len(mappings) != 1:
            return None
        (result,) = mappings
        return result

    @property
    def fiat_equivalent(self):
        return self._element

    def _is_simplex(self):
        return self.complex.is_simplex()

    def basis_evaluation(self, order, ps, entity=None):
        """{alpha: Table} at the point set ``ps`` on the reference element
        (finat/fiat_elements.py:60-123): derivative == degree on a simplex -> asserted constant and
        stored without the point axis, derivative > degree -> asserted zero, else pointwise."""
        fiat_result = self._element.tabulate(order, ps.points, entity)
        index_shape, value_shape = self.index_shape, tuple(self.value_shape)
        result = {}
        for alpha, fiat_table in fiat_result.items():
            derivative = sum(alpha)
            if derivative == self.degree and self._is_simplex():
                fiat_table = fiat_table.reshape(*index_shape, *value_shape, -1)
                assert numpy.allclose(fiat_table, fiat_table[..., 0, None])
                result[alpha] = Table(numpy.ascontiguousarray(fiat_table[..., 0]), CELLWISE_CONSTANT)
            elif derivative > self.degree:
                assert numpy.allclose(fiat_table, 0.0)
                result[alpha] = Table(numpy.zeros(index_shape + value_shape), ZERO)
            else:
                result[alpha] = Table(fiat_table.reshape(index_shape + value_shape + ps.extents), POINTWISE)
        return result

    @property
    def dual_basis(self):
        """(Q, PointSet): the dual basis as a weight tensor over the unique evaluation points,
        ``dof_i(f) = sum_{k, cmp} Q[i, k, *cmp] f(x_k)[cmp]`` (finat/fiat_elements.py:162-258 without GEM:
        Q is the dense array that would become the GEM literal; ``Q_is_identity`` tells when FInAT would
        express it symbolically as a Kronecker delta).  Point order as in the reference: sorted points of
        each functional in order of first appearance, duplicates (atol 1e-12) merged."""
        if getattr(self, "_dual_cache", None) is None:
            nodes = self._element.dual_basis()[:self.space_dimension()]
            seen, allpts = {}, []
            for dual in nodes:
                if len(dual.deriv_dict) != 0:
                    raise NotImplementedError("FIAT dual bases with derivative nodes represented via a ``Functional.deriv_dict`` "
                                              "property do not currently have a FInAT dual basis")
                pts = tuple(sorted(dual.get_point_dict().keys()))
                if pts not in seen:
                    seen[pts] = (len(allpts), len(allpts) + len(pts))
                    allpts.extend(pts)
            unique, index = [], []
            for p in allpts:
                for j in reversed(range(len(unique))):
                    if numpy.allclose(unique[j], p, atol=1e-12):
                        index.append(j)
                        break
                else:
                    index.append(len(unique))
                    unique.append(p)
            entries = {}
            for i, dual in enumerate(nodes):
                point_dict = dual.get_point_dict()
                pts = tuple(sorted(point_dict.keys()))
                k0, k1 = seen[pts]
                for p, k in zip(pts, index[k0:k1]):
                    for weight, cmp in point_dict[p]:
                        entries[(i, k, *cmp)] = weight
            shape = tuple(m + 1 for m in map(max, zip(*entries)))
            Q = numpy.zeros(shape)
            for idx, value in entries.items():
                Q[idx] = value
            self.Q_is_identity = all(len(key) == 2 and len(set(key)) == 1 and numpy.isclose(w, 1) for key, w in entries.items())
            self._dual_cache = (Q, PointSet(numpy.asarray(unique)))
        return self._dual_cache

    def dual_evaluation_batch(self, values, stream=None):
        """Degrees of freedom of a batch of functions given by their values at the dual basis' points:
        ``values`` (ncell, npts, *value_shape) -> device tensor (ncell, ndof), ``dofs = Q : values`` on the
        device (fx_riesz_assemble); interpolation into the element's space when ``values`` are function
        values at ``dual_basis[1].points``."""
        import torch
        Q, ps = self.dual_basis
        ctx = runtime.Context.get()
        values = runtime._as_device(values, ctx)
        ncell = values.shape[0]
        if tuple(values.shape[1:]) != tuple(Q.shape[1:]):
            raise ValueError(f"values must have shape (ncell, {', '.join(map(str, Q.shape[1:]))})")
        Qd = torch.as_tensor(Q.reshape(Q.shape[0], -1)).to(ctx.device)
        return runtime.riesz_assemble(values.reshape(ncell, -1), Qd, ctx)

    def basis_evaluation_batch(self, order, points, verts=None, pushforward=False, check=True, stream=None):
        """Batched ``basis_evaluation``: ``points`` (nreq, npts, sd) [+ per-request cells ``verts``]
        -> {alpha: Table} whose arrays are views of ONE device tensor (nreq, ntab, ndof, *value_shape, npts)
        — no copy, nothing leaves the GPU.  The constant / zero facts FInAT asserts are evaluated on
        the device for every request (fx_classify_tables) when ``check`` is true: a table that violates
        them raises AssertionError, as the reference does; cellwise-constant tables are returned as
        their first point column (one value per request and basis function)."""
        el = self._element
        tabs = el.tabulate_batch(order, points, verts=verts, stream=stream, pushforward=pushforward)
        sd = self.cell.get_spatial_dimension()
        alphas = [a for k in range(order + 1) for a in mis(sd, k)]
        nreq, ntab = tabs.shape[:2]
        npts = tabs.shape[-1]
        rows = int(numpy.prod(tabs.shape[2:-1], dtype=numpy.int64))
        kinds = []
        for alpha in alphas:
            derivative = sum(alpha)
            kinds.append(CELLWISE_CONSTANT if derivative == self.degree and self._is_simplex()
                         else ZERO if derivative > self.degree else POINTWISE)
        if check and nreq and any(k != POINTWISE for k in kinds):
            stats = runtime.classify_tables(tabs.reshape(nreq, ntab, rows, npts), rtol=_RTOL, stream=stream)
            worst = stats.amax(dim=0).cpu().numpy()  # NaN propagates
            for t, kind in enumerate(kinds):
                if kind == ZERO:
                    assert worst[t, 0] <= _ATOL, f"table {alphas[t]} should vanish, max |x| = {worst[t, 0]}"
                elif kind == CELLWISE_CONSTANT:
                    assert worst[t, 1] <= _ATOL, f"table {alphas[t]} should be constant on each cell"
        result = {}
        for t, (alpha, kind) in enumerate(zip(alphas, kinds)):
            if kind == POINTWISE:
                result[alpha] = Table(tabs[:, t], kind)
            elif kind == CELLWISE_CONSTANT:
                result[alpha] = Table(tabs[:, t, ..., 0], kind)
            else:
                result[alpha] = Table(tabs.new_zeros(tuple(tabs.shape[2:-1])), kind)
        return result


class TableArgument:
    """A tabulation passed to the generated kernel as an argument: its name and shape
    (``gem.Variable(name, shape)`` in finat/runtime_tabulated.py:80-94)."""

    def __init__(self, name, shape):
        self.name = name
        self.shape = tuple(shape)

    def __repr__(self):
        return f"TableArgument({self.name!r}, {self.shape})"


class RuntimeTabulated:
    """finat/runtime_tabulated.py:10-95: a 1-D element whose tables arrive at run time.
    ``basis_evaluation`` names the arguments exactly as the reference does;
    ``tabulate_arguments`` (added) fills them on the device for a batch of point sets from a 1-D
    element of this package, in the argument layout (points, basis functions)."""

    def __init__(self, cell, degree, variant=None, shift_axes=0, restriction=None, continuous=True):
        if cell.get_spatial_dimension() != 1:
            raise NotImplementedError("Runtime tabulated elements limited to 1D.")
        assert isinstance(variant, str)
        assert isinstance(shift_axes, int) and 0 <= shift_axes
        assert isinstance(continuous, bool)
        assert restriction in [None, '+', '-']
        self.cell = cell
        self.degree = degree
        self.variant = variant
        self.shift_axes = shift_axes
        self.restriction = restriction
        self.continuous = continuous

    @property
    def formdegree(self):
        return 0 if self.continuous else self.cell.get_spatial_dimension()

    def entity_dofs(self):
        raise NotImplementedError("I cannot tell where my DoFs are... :-/")

    def space_dimension(self):
        return self.degree + 1

    @property
    def index_shape(self):
        return (self.space_dimension(),)

    @property
    def value_shape(self):
        return ()

    @property
    def mapping(self):
        return "affine"

    def argument_name(self, alpha):
        return str.format("rt_{}_{}_{}_{}_{}_{}", self.variant, self.degree, ''.join(map(str, alpha)), self.shift_axes,
                          'c' if self.continuous else 'd', {None: "", '+': "p", '-': "m"}[self.restriction])

    def basis_evaluation(self, order, ps, entity=None):
        shape = ps.extents + self.index_shape + self.value_shape
        return {alpha: TableArgument(self.argument_name(alpha), shape)
                for derivative in range(order + 1) for alpha in mis(1, derivative)}

    def point_evaluation(self, order, point, entity=None):
        raise NotImplementedError("Point evaluation not supported for runtime tabulated elements")

    def tabulate_arguments(self, order, points, fiat_element, stream=None):
        """{argument name: device tensor (nreq, npts, degree + 1)} for the point sets ``points``
        (nreq, npts) on the interval, tabulated from ``fiat_element`` (a 1-D nodal element of this
        package with degree + 1 basis functions) and transposed on the device to the
        kernel-argument layout (fx_line_tabulate_batch + fx_tables_point_major)."""
        if fiat_element.space_dimension() != self.space_dimension():
            raise ValueError("the element providing the tables must have degree + 1 basis functions")
        es = fiat_element.get_nodal_basis().get_expansion_set()
        if not hasattr(es, "device_line"):
            raise NotImplementedError("run-time tables need a 1-D Lagrange-type element (barycentric_interpolation)")
        line = es.device_line()
        tabs = line.tabulate_batch(order, points, stream=stream)  # (nreq, order + 1, nn, npts)
        out = {}
        for k in range(order + 1):
            out[self.argument_name((k,))] = runtime.tables_point_major(tabs[:, k], stream=stream)
        return out


class TensorProductElement:
    """Per-factor tables for a sum-factorised evaluation (finat/tensor_product.py:103-144): the
    product over factors is formed symbolically by the consumer and never materialised, so what the
    tabulator owes it is each factor's tables at that factor's points and, per multi-index of the
    product, which factor derivative enters."""

    def __init__(self, factors):
        self.factors = tuple(factors)  # FiatElement / RuntimeTabulated adapters

    @property
    def index_shape(self):
        return tuple(n for fe in self.factors for n in fe.index_shape)

    def factor_multiindices(self, order):
        """{Delta: (delta_factor_0, delta_factor_1, ...)} for all ``Delta in mis(dimension, k)``, k <= order
        (finat/tensor_product.py:109-124)."""
        dims = [fe.cell.get_spatial_dimension() for fe in self.factors]
        bounds = numpy.concatenate([[0], numpy.cumsum(dims)])
        result = {}
        for derivative in range(order + 1):
            for Delta in mis(int(bounds[-1]), derivative):
                result[Delta] = tuple(tuple(Delta[bounds[i]:bounds[i + 1]]) for i in range(len(dims)))
        return result

    def basis_evaluation(self, order, ps_factors, entity=None):
        """(factor results, {Delta: factor multi-indices}): ``factor results[i]`` is
        ``factors[i].basis_evaluation(order, ps_factors[i])``."""
        assert len(ps_factors) == len(self.factors)
        factor_results = [fe.basis_evaluation(order, ps_) for fe, ps_ in zip(self.factors, ps_factors)]
        return factor_results, self.factor_multiindices(order)
