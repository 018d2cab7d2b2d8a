"""ExpansionSet facade: the orthonormal Dubiner basis (and its "bubble"/"dual"
variants) on a simplex, tabulated by the HIP kernel.

Same constructor and methods as FIAT/expansions.py ExpansionSet (:342-635):
``ExpansionSet(ref_el, scale=None, variant=None)``, ``_tabulate(n, pts, order)``
-> {alpha: (nexp, npts)}, ``tabulate(n, pts)``, ``get_num_members``,
``get_scale``.  Every evaluation runs on the GPU; there is no CPU path.
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
    if continuity == "C0":
        top = ref_el.get_topology()
        return sum(math.comb(n - 1, dim) * len(top[dim]) for dim in top)
    return math.comb(n + sd, sd)


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


class ExpansionSet:
    def __init__(self, ref_el, scale=None, variant=None):
        if variant not in (None, "bubble", "dual"):
            raise ValueError(f"Invalid variant {variant}")
        sd = ref_el.get_spatial_dimension()
        if ref_el.get_shape() not in (reference_element.LINE, reference_element.TRIANGLE,
                                      reference_element.TETRAHEDRON):
            raise ValueError("Invalid reference element type.")
        self.ref_el = ref_el
        self.variant = variant
        if scale is None:
            scale = math.sqrt(1.0 / reference_element.default_simplex(sd).volume())
        elif isinstance(scale, str):
            vol = ref_el.volume()
            key = scale.lower()
            if key == "orthonormal":
                scale = math.sqrt(1.0 / vol)
            elif key == "l2 piola":
                scale = 1.0 / vol
            else:
                raise ValueError(f"Invalid scale {scale}")
        self.scale = scale
        self.continuity = "C0" if variant == "bubble" else None
        self.recurrence_order = math.inf
        self._dev = {}

    def get_scale(self, n, cell=0):
        sd = self.ref_el.get_spatial_dimension()
        if n == 0 and sd > 1:
            return 1
        return self.scale

    def get_num_members(self, n):
        return polynomial_dimension(self.ref_el, n, self.continuity)

    def _device_set(self, n):
        """Identity-coefficient polynomial set on the device (cached per degree)."""
        if n not in self._dev:
            sd = self.ref_el.get_spatial_dimension()
            self._dev[n] = runtime.SimplexPolySet(sd, n, variant=self.variant, scale=self.get_scale(n),
                                                  verts=numpy.asarray(self.ref_el.get_vertices()))
        return self._dev[n]

    def _tabulate(self, n, pts, order=0):
        """{alpha: table[i, j] = D^alpha phi_i(pts[j])}; a single point drops the last axis."""
        pts = numpy.asarray(pts, dtype=float)
        sd = self.ref_el.get_spatial_dimension()
        single = pts.ndim == 1
        P = pts.reshape(1, -1, sd)
        out = self._device_set(n).tabulate_batch(order, P).cpu().numpy()[0]
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
