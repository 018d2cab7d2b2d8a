"""ExpansionSet facade: the orthonormal Dubiner basis (and its "bubble"/"dual"
variants) on a simplex, tabulated by the HIP kernel.

Same constructor and methods as FIAT/expansions.py ExpansionSet (:342-635):
``ExpansionSet(ref_el, scale=None, variant=None)``, ``_tabulate(n, pts, order)``
-> {alpha: (nexp, npts)}, ``tabulate(n, pts)``, ``get_num_members``,
``get_scale``.  Every evaluation runs on the GPU; there is no CPU path.

On a macro cell (``ref_el.is_macrocell()``, macro.py) the set is the piecewise expansion set of
FIAT/expansions.py:449-490: members numbered by ``polynomial_entity_ids`` over the complex,
``get_cell_node_map`` (:744-768) maps the members of each sub-cell into them; binning, per-cell
recurrence and scatter run in the MACRO instance of the generic kernel (``fx_macro_tabulate_batch``).
"""
import math

import numpy

from . import reference_element, runtime
from .polynomial_set_util import mis


def morton_index2(p, q=0):
    return (p + q) * (p + q + 1) // 2 + q


def morton_index3(p, q=0, r=0):
    return (p + q + r) * (p + q + r + 1) * (p + q + r + 2) // 6 + (q + r) * (q + r + 1) // 2 + r


def polynomial_dimension(ref_el, n, continuity=None):
    """Dimension of P_n on the cell (C0 numbering gives the same count for n >= 1)."""
    if ref_el.get_shape() == reference_element.POINT:
        if n > 0:
            raise ValueError("Only degree zero polynomials supported on point elements.")
        return 1
    sd = ref_el.get_spatial_dimension()
    top = ref_el.get_topology()
    if continuity == "C0":
        return sum(math.comb(n - 1, dim) * len(top[dim]) for dim in top)
    return math.comb(n + sd, sd) * len(top[sd])


def polynomial_cell_node_map(ref_el, n, continuity=None):
    """(ncell, nexp) members of the complex carried by each cell, in the member order of a single
    cell (FIAT/expansions.py:744-768)."""
    top = ref_el.get_topology()
    sd = ref_el.get_spatial_dimension()
    entity_ids = polynomial_entity_ids(ref_el, n, continuity)
    ref_ids = polynomial_entity_ids(ref_el.construct_subelement(sd), n, continuity)
    width = sum(len(ref_ids[dim][e]) for dim in ref_ids for e in ref_ids[dim])
    cmap = numpy.zeros((len(top[sd]), width), dtype=int)
    if len(top[sd]) == 1:
        conn = {0: {dim: sorted(top[dim]) for dim in top}}
    else:
        conn = ref_el.get_cell_connectivity()
    for cell in sorted(top[sd]):
        for dim in top:
            for ref_entity, entity in enumerate(conn[cell][dim]):
                cmap[cell, ref_ids[dim][ref_entity]] = entity_ids[dim][entity]
    return cmap


def polynomial_entity_ids(ref_el, n, continuity=None):
    """{dim: {entity: members}} of the hierarchical numbering of a degree-n expansion set
    (FIAT/expansions.py:717-741): C0 sets own comb(n - 1, dim) members per entity of dimension dim,
    discontinuous sets keep everything in the cell."""
    top = ref_el.get_topology()
    sd = ref_el.get_spatial_dimension()
    entity_ids, cur = {}, 0
    for dim in sorted(top):
        dofs = math.comb(n - 1, dim) if continuity == "C0" else (math.comb(n + dim, dim) if dim == sd else 0)
        entity_ids[dim] = {}
        for entity in sorted(top[dim]):
            entity_ids[dim][entity] = list(range(cur, cur + dofs))
            cur += dofs
    return entity_ids


def compute_cell_point_map(ref_el, pts, unique=True, tol=1e-12):
    """{cell: indices of the points in that cell} on a complex (FIAT/expansions.py:771-811), host version for
    the handful of points used while C^k spaces are constructed (tabulate_jumps); batched tabulation bins on
    the device (macro_small.hpp, simplex_kernel.hpp)."""
    sd = ref_el.get_spatial_dimension()
    top = ref_el.get_topology()
    pts = numpy.asarray(pts, dtype=float).reshape(-1, sd)
    if len(top[sd]) == 1:
        return {0: list(range(len(pts)))}

    def distance(verts):
        A, b = reference_element.make_affine_mapping(verts, numpy.eye(sd + 1))
        h = 1.0 / numpy.linalg.norm(A, axis=1)
        bary = pts @ (A * h[:, None]).T + b * h
        return 0.5 * numpy.abs(numpy.sum(numpy.abs(bary) - bary, axis=-1))

    limit = distance(ref_el.get_parent().get_vertices()) + tol
    taken = numpy.zeros(len(pts), dtype=bool)
    result = {}
    for cell in sorted(top[sd]):
        near = distance(ref_el.get_vertices_of_subcomplex(top[sd][cell])) < limit
        if unique:
            near &= ~taken
            taken |= near
        if near.any():
            result[cell] = [int(i) for i in numpy.where(near)[0]]
    return result


class ExpansionSet:
    def __init__(self, ref_el, scale=None, variant=None):
        if variant not in (None, "bubble", "dual"):
            raise ValueError(f"Invalid variant {variant}")
        sd = ref_el.get_spatial_dimension()
        if ref_el.get_shape() not in (reference_element.POINT, reference_element.LINE, reference_element.TRIANGLE,
                                      reference_element.TETRAHEDRON):
            raise ValueError("Invalid reference element type.")
        self.ref_el = ref_el
        self.variant = variant
        self.num_cells = len(ref_el.get_topology()[sd])
        if scale is None:
            scale = 1.0 if sd == 0 else math.sqrt(1.0 / reference_element.default_simplex(sd).volume())
        elif isinstance(scale, str):
            if scale.lower() not in ("orthonormal", "l2 piola"):
                raise ValueError(f"Invalid scale {scale}")
            if self.num_cells == 1:
                scale = self._named_scale(scale, ref_el.volume())
        self.scale = scale
        self.continuity = "C0" if variant == "bubble" else None
        self.recurrence_order = math.inf
        self._dev = {}
        self._cell_node_map_cache = {}

    @staticmethod
    def _named_scale(name, vol):
        return math.sqrt(1.0 / vol) if name.lower() == "orthonormal" else 1.0 / vol

    def get_scale(self, n, cell=0):
        """FIAT/expansions.py:386-399."""
        sd = self.ref_el.get_spatial_dimension()
        if isinstance(self.scale, str):
            return self._named_scale(self.scale, self.ref_el.volume_of_subcomplex(sd, cell))
        if n == 0 and sd > 1 and self.num_cells == 1:
            return 1
        return self.scale

    def get_cell_node_map(self, n):
        if n not in self._cell_node_map_cache:
            self._cell_node_map_cache[n] = polynomial_cell_node_map(self.ref_el, n, self.continuity)
        return self._cell_node_map_cache[n]

    def device_polyset(self, n, coeffs=None, value_shape=()):
        """Device polynomial set over this expansion set: fx_element on a single cell, fx_macro_element
        on a complex."""
        sd = self.ref_el.get_spatial_dimension()
        if self.num_cells == 1:
            return runtime.SimplexPolySet(sd, n, variant=self.variant, scale=self.get_scale(n),
                                          verts=numpy.asarray(self.ref_el.get_vertices()), coeffs=coeffs,
                                          value_shape=value_shape)
        top = self.ref_el.get_topology()
        cells = numpy.array([self.ref_el.get_vertices_of_subcomplex(top[sd][c]) for c in sorted(top[sd])])
        scales = numpy.array([float(self.get_scale(n, c)) for c in sorted(top[sd])])
        return runtime.MacroPolySet(sd, n, self.variant, scales[0], numpy.asarray(self.ref_el.get_parent().get_vertices()),
                                    cells, self.get_cell_node_map(n), self.get_num_members(n),
                                    cell_scale=scales / scales[0], coeffs=coeffs, value_shape=value_shape)

    def get_num_members(self, n):
        return polynomial_dimension(self.ref_el, n, self.continuity)

    def _device_set(self, n):
        """Identity-coefficient polynomial set on the device (cached per degree)."""
        if n not in self._dev:
            self._dev[n] = self.device_polyset(n)
        return self._dev[n]

    def _tabulate_on_cell(self, n, pts, order=0, cell=0):
        """{alpha: (nexp, npts)} of the polynomials of ONE sub-cell, wherever the points lie
        (FIAT/expansions.py:411-447) -- the single-cell kernels on the sub-cell's vertices."""
        key = ("cell", n, cell)
        if key not in self._dev:
            sd = self.ref_el.get_spatial_dimension()
            top = self.ref_el.get_topology()
            self._dev[key] = runtime.SimplexPolySet(
                sd, n, variant=self.variant, scale=float(self.get_scale(n, cell)),
                verts=numpy.asarray(self.ref_el.get_vertices_of_subcomplex(top[sd][cell])))
        sd = self.ref_el.get_spatial_dimension()
        P = numpy.asarray(pts, dtype=float).reshape(1, -1, sd)
        out = runtime.fetch(self._dev[key].tabulate_batch(order, P))[0]
        keys = [a for k in range(order + 1) for a in mis(sd, k)]
        return {a: out[t] for t, a in enumerate(keys)}

    def get_dmats(self, degree, cell=0):
        """dmats[d][i, j]: d/dx_d phi_j = sum_i dmats[d][i, j] phi_i on one sub-cell -- the array FIAT/expansions.py:576-600
        returns (solve(V^T, dV^T); the regression suite's reference-polynomials data pin this orientation) -- from a device
        tabulation of values and gradients at a unisolvent interior lattice."""
        key = ("dmats", degree, cell)
        if key not in self._dev:
            sd = self.ref_el.get_spatial_dimension()
            top = self.ref_el.get_topology()
            verts = self.ref_el.get_vertices_of_subcomplex(top[sd][cell])
            lattice = reference_element.make_lattice(verts, degree, variant="equispaced_interior") if degree > 0 \
                else [tuple(numpy.mean(numpy.asarray(verts), axis=0))]
            tab = self._tabulate_on_cell(degree, numpy.array(lattice), 1, cell=cell)
            V = tab[(0,) * sd]
            self._dev[key] = [numpy.linalg.solve(V.T, tab[alpha].T) for alpha in mis(sd, 1)]
        return self._dev[key]

    def _tabulate(self, n, pts, order=0):
        """{alpha: table[i, j] = D^alpha phi_i(pts[j])}; a single point drops the last axis."""
        sd = self.ref_el.get_spatial_dimension()
        if sd == 0:     # the one constant on a point (FIAT/expansions.py:638-649): host arithmetic, nothing to launch
            if n != 0 or order != 0:
                raise ValueError("Only degree zero polynomials and no derivatives on point elements.")
            return {(): numpy.ones((1, len(pts)))}
        pts = numpy.asarray(pts, dtype=float)
        single = pts.ndim == 1
        P = pts.reshape(1, -1, sd)
        out = runtime.fetch(self._device_set(n).tabulate_batch(order, P))[0]
        keys = [a for k in range(order + 1) for a in mis(sd, k)]
        result = {a: numpy.ascontiguousarray(out[t]) for t, a in enumerate(keys)}
        if single:
            result = {a: v[..., 0] for a, v in result.items()}
        return result

    def tabulate(self, n, pts):
        if len(pts) == 0:
            return numpy.array([])
        sd = self.ref_el.get_spatial_dimension()
        return self._tabulate(n, pts)[(0,) * sd]

    def tabulate_derivatives(self, n, pts):
        sd = self.ref_el.get_spatial_dimension()
        vals = self._tabulate(n, pts, order=1)
        v = vals[(0,) * sd]
        dv = [vals[alpha] for alpha in mis(sd, 1)]
        return [[(v[i, j], [vi[i, j] for vi in dv]) for j in range(v.shape[1])] for i in range(v.shape[0])]

    def tabulate_jet(self, n, pts, order=1):
        sd = self.ref_el.get_spatial_dimension()
        vals = self._tabulate(n, pts, order=order)
        v0 = vals[(0,) * sd]
        data = [v0]
        for r in range(1, order + 1):
            vr = numpy.zeros((sd,) * r + v0.shape, dtype=v0.dtype)
            for index in numpy.ndindex(vr.shape[:r]):
                vr[index] = vals[tuple(map(index.count, range(sd)))]
            data.append(vr.transpose((r, r + 1) + tuple(range(r))))
        return data

    def __eq__(self, other):
        return (type(self) is type(other) and self.ref_el == other.ref_el
                and self.continuity == other.continuity)

    def __hash__(self):
        return hash((type(self).__name__, self.ref_el, self.continuity))


# The reference's per-cell class names (FIAT/expansions.py:638-716: ExpansionSet.__new__ dispatches on the cell shape; here one
# class serves every simplex, the named ones check the dimension they are asked for)
def _dimension_checked(name, dim, what):
    class _Named(ExpansionSet):
        def __init__(self, ref_el, **kwargs):
            if ref_el.get_spatial_dimension() != dim:
                raise ValueError(f"Must have a {what}")
            super().__init__(ref_el, **kwargs)
    _Named.__name__ = _Named.__qualname__ = name
    return _Named


LineExpansionSet = _dimension_checked("LineExpansionSet", 1, "line")
TriangleExpansionSet = _dimension_checked("TriangleExpansionSet", 2, "triangle")
TetrahedronExpansionSet = _dimension_checked("TetrahedronExpansionSet", 3, "tetrahedron")
