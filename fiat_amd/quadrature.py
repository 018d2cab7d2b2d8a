"""Quadrature rules that *produce the points* fed to tabulate() and the weights of
integral-moment degrees of freedom.  Host-side input producers.

Mirrors FIAT/quadrature.py (QuadratureRule :47-93, GaussJacobiQuadratureLineRule
:96-110, CollapsedQuadratureSimplexRule :171-181, FacetQuadratureRule :198-224,
make_quadrature :227-255, make_tensor_product_quadrature :258-268) and
FIAT/quadrature_schemes.py create_quadrature (:46-106): the "default" scheme serves the
tabulated Xiao-Gimbutas / classical rules on triangles and tetrahedra (data:
fiat_amd/data/simplex_rules.npz, tools/make_rule_tables.py), "canonical" the collapsed
Gauss-Jacobi rule; ``QuadratureRule.device_points()`` keeps a rule resident on the GPU for
``element.tabulate_cells``.
"""
import itertools

import numpy
from scipy.special import roots_jacobi

from . import reference_element


def pseudo_determinant(A):
    return numpy.sqrt(abs(numpy.linalg.det(numpy.dot(A.T, A))))


def map_quadrature(pts_ref, wts_ref, source_cell, target_cell, jacobian=False, avg=False):
    """Push a rule from source_cell to target_cell (affine); with avg=True the
    weights are kept (averages instead of integrals)."""
    A, b = reference_element.make_affine_mapping(source_cell.get_vertices(), target_cell.get_vertices())
    pts_ref = numpy.asarray(pts_ref, dtype=float)
    if pts_ref.ndim != 2:
        pts_ref = pts_ref.reshape(-1, A.shape[1])
    pts = pts_ref @ A.T + b
    wts = numpy.asarray(wts_ref, dtype=float)
    if not avg:
        wts = wts * pseudo_determinant(A)
    pts = tuple(map(tuple, pts))
    wts = tuple(wts.flat)
    return (pts, wts, A) if jacobian else (pts, wts)


class QuadratureRule:
    def __init__(self, ref_el, pts, wts):
        if len(wts) != len(pts):
            raise ValueError("Have %d weights, but %d points" % (len(wts), len(pts)))
        self.ref_el = ref_el
        self.pts = pts
        self.wts = wts

    def get_points(self):
        return numpy.array(self.pts)

    def get_weights(self):
        return numpy.array(self.wts)

    def device_points(self):
        """(points, weights) as device tensors, uploaded once per rule: the input of ``element.tabulate_cells``."""
        if getattr(self, "_device", None) is None:
            from . import runtime
            ctx = runtime.Context.get()
            self._device = (runtime._as_device(self.get_points(), ctx), runtime._as_device(self.get_weights(), ctx))
        return self._device

    def integrate(self, f):
        return sum(w * f(x) for x, w in zip(self.pts, self.wts))


class GaussJacobiQuadratureLineRule(QuadratureRule):
    """m-point Gauss-Jacobi rule with weight (1-x)^a (1+x)^b mapped to ref_el."""

    def __init__(self, ref_el, m, a=0, b=0):
        x, w = roots_jacobi(m, a, b)
        pts, wts = map_quadrature(x.reshape(-1, 1), w, reference_element.default_simplex(1), ref_el)
        super().__init__(ref_el, pts, wts)


class GaussLegendreQuadratureLineRule(GaussJacobiQuadratureLineRule):
    def __init__(self, ref_el, m):
        super().__init__(ref_el, m)


class GaussLobattoLegendreQuadratureLineRule(QuadratureRule):
    """m-point Gauss-Lobatto-Legendre rule on the interval, exact to degree 2m - 3 (FIAT/quadrature.py:113-125): the end
    points and the roots of P'_{m-1} (Jacobi(1, 1) of degree m - 2), weights 2 / (m (m - 1) P_{m-1}(x)^2)."""

    def __init__(self, ref_el, m):
        if m < 2:
            raise ValueError("Gauss-Labotto-Legendre quadrature invalid for fewer than 2 points")
        from scipy.special import eval_legendre
        inner = roots_jacobi(m - 2, 1, 1)[0] if m > 2 else numpy.zeros(0)
        x = numpy.concatenate([[-1.0], inner, [1.0]])
        w = 2.0 / (m * (m - 1) * eval_legendre(m - 1, x) ** 2)
        pts, wts = map_quadrature(x.reshape(-1, 1), w, reference_element.default_simplex(1), ref_el)
        super().__init__(ref_el, pts, wts)


class RadauQuadratureLineRule(QuadratureRule):
    """m-point Gauss-Radau rule on the interval with the right (default) or left end point as a node, exact to degree 2m - 2
    (FIAT/quadrature.py:137-168): the interior nodes are Gauss-Jacobi nodes for the weight that vanishes at the fixed end,
    with that weight divided out of their weights; the end point takes what is left of the interval's length."""

    def __init__(self, ref_el, m, right=True):
        if m < 1:
            raise ValueError("Gauss-Radau quadrature invalid for fewer than 1 points")
        end = 1 if right else 0
        x0 = tuple(float(c) for c in ref_el.get_vertices()[end])
        length = ref_el.volume()
        pts, wts = (), ()
        if m > 1:
            inner = GaussJacobiQuadratureLineRule(ref_el, m - 1, end, 1 - end)
            x = inner.get_points().reshape(-1)
            wts = tuple(inner.get_weights() / ((2.0 / length) * numpy.abs(x0[0] - x)))
            pts = tuple(inner.pts)
        w0 = length - sum(wts)
        super().__init__(ref_el, (*pts, x0) if right else (x0, *pts), (*wts, w0) if right else (w0, *wts))


def _collapsed_rule(dim, m):
    """Product of Gauss-Jacobi(j, 0) rules on the cube mapped by the Duffy
    transformation onto the (-1,1)^dim simplex (Karniadakis & Sherwin)."""
    rules = [roots_jacobi(m, j, 0) for j in range(dim)]
    pts, wts = [], []
    for idx in itertools.product(range(m), repeat=dim):
        e = [rules[j][0][idx[j]] for j in range(dim)]
        w = 1.0
        for j in range(dim):
            w *= rules[j][1][idx[j]] / 2.0 ** j
        x = []
        for i in range(dim):
            f = 1.0 + e[i]
            for j in range(i + 1, dim):
                f *= (1.0 - e[j]) / 2.0
            x.append(f - 1.0)
        pts.append(x)
        wts.append(w)
    return numpy.array(pts), numpy.array(wts)


class CollapsedQuadratureSimplexRule(QuadratureRule):
    def __init__(self, ref_el, m):
        dim = ref_el.get_spatial_dimension()
        pts_ref, wts_ref = _collapsed_rule(dim, m)
        pts, wts = map_quadrature(pts_ref, wts_ref, reference_element.default_simplex(dim), ref_el)
        super().__init__(ref_el, pts, wts)


class FacetQuadratureRule(QuadratureRule):
    """A rule on sub-entity (entity_dim, entity_id) of ref_el mapped from a rule on
    the reference sub-entity."""

    def __init__(self, ref_el, entity_dim, entity_id, Q_ref, avg=False):
        facet = ref_el.construct_subelement(entity_dim)
        verts = ref_el.get_vertices_of_subcomplex(ref_el.get_topology()[entity_dim][entity_id])
        facet = reference_element.UFCSimplex(facet.get_shape(), verts, facet.get_topology()) \
            if entity_dim > 0 else facet
        if entity_dim == 0:     # a vertex: the one-point rule sits at the vertex's coordinates in the cell (no Jacobian to speak of)
            wts = tuple(float(w) for w in Q_ref.get_weights())
            pts = (tuple(float(c) for c in verts[0]),) * len(wts)
            J = numpy.zeros((len(verts[0]), 0))
        else:
            pts, wts, J = map_quadrature(Q_ref.get_points(), Q_ref.get_weights(), Q_ref.ref_el, facet,
                                         jacobian=True, avg=avg)
        super().__init__(facet, pts, wts)
        self._J = J
        self._reference_rule = Q_ref

    def reference_rule(self):
        return self._reference_rule

    def jacobian(self):
        return self._J

    def jacobian_determinant(self):
        return pseudo_determinant(self._J)


def make_quadrature(ref_el, m):
    """Collapsed rule with m points per direction."""
    shape = ref_el.get_shape()
    if shape == reference_element.POINT:
        return QuadratureRule(ref_el, [()], [1])
    if shape == reference_element.LINE:
        return GaussJacobiQuadratureLineRule(ref_el, m)
    if shape in (reference_element.TRIANGLE, reference_element.TETRAHEDRON):
        return CollapsedQuadratureSimplexRule(ref_el, m)
    raise ValueError("Unable to make quadrature for cell: %s" % ref_el)


def make_tensor_product_quadrature(*quad_rules):
    """Points concatenated, weights multiplied; the last factor varies fastest."""
    ref_el = reference_element.TensorProductCell(*[q.ref_el for q in quad_rules])
    pts = [list(itertools.chain(*pt)) for pt in itertools.product(*[q.pts for q in quad_rules])]
    wts = [numpy.prod(wt) for wt in itertools.product(*[q.wts for q in quad_rules])]
    return QuadratureRule(ref_el, pts, wts)


_TABLES = None


def _rule_tables():
    """fiat_amd/data/simplex_rules.npz: the Xiao-Gimbutas tables (symmetric simplex) and the classical low-degree
    rules (UFC simplex) behind the reference's default scheme; plain arrays produced by tools/make_rule_tables.py."""
    global _TABLES
    if _TABLES is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "simplex_rules.npz")
        with numpy.load(path) as data:
            _TABLES = {k: data[k] for k in data.files}
    return _TABLES


def tabulated_rule(sd, degree):
    """(cell vertices, points, weights) of the tabulated rule the default scheme uses for this dimension and
    degree, or None when it falls back to the collapsed Gauss-Jacobi rule (FIAT/quadrature_schemes.py:356-419:
    classical rules for triangle degree <= 3 / tetrahedron degree <= 2, Xiao-Gimbutas up to 50 / 15)."""
    if sd not in (2, 3):
        return None
    T = _rule_tables()
    degree = max(int(degree), 0)
    for family in ("classical", "xg"):
        degrees = T[f"{family}{sd}_degrees"]
        hit = numpy.nonzero(degrees == degree)[0]
        if len(hit):
            lo, hi = T[f"{family}{sd}_offsets"][hit[0]:hit[0] + 2]
            return T[f"{family}{sd}_cell"], T[f"{family}{sd}_points"][lo:hi], T[f"{family}{sd}_weights"][lo:hi]
    return None


class _TabulatedSimplexRule(QuadratureRule):
    def __init__(self, ref_el, source_verts, pts, wts):
        source = reference_element.physical_simplex(source_verts)
        super().__init__(ref_el, *map_quadrature(pts, wts, source, ref_el))


def create_quadrature(ref_el, degree, scheme="default", entity=None):
    """Rule exact for polynomials of the given degree on ref_el, or on one of its sub-entities; the same
    points and weights as FIAT/quadrature_schemes.py:46-106 for "default" and "canonical"."""
    if entity is not None:
        dim, number = entity
        return FacetQuadratureRule(ref_el, dim, number, create_quadrature(ref_el.construct_subelement(dim), degree, scheme=scheme))
    if ref_el.is_macrocell():     # composite rule that respects the splitting
        from .macro import MacroQuadratureRule
        sub = ref_el.construct_subelement(ref_el.get_spatial_dimension())
        return MacroQuadratureRule(ref_el, create_quadrature(sub, degree, scheme=scheme))
    if isinstance(ref_el, reference_element.TensorProductCell):
        degrees = tuple(degree) if numpy.ndim(degree) else (degree,) * len(ref_el.cells)
        assert len(degrees) == len(ref_el.cells)
        return make_tensor_product_quadrature(*(create_quadrature(c, d, scheme) for c, d in zip(ref_el.cells, degrees)))
    if isinstance(ref_el, reference_element.Hypercube):     # the rule of the product of intervals it flattens
        return create_quadrature(ref_el.product, degree, scheme)
    if degree < 0:
        raise ValueError("Need positive degree, not %d" % degree)
    if scheme == "KMV":
        raise NotImplementedError("the Kong-Mulder-Veldhuizen lumped rules are out of scope for fiat_amd")
    if scheme not in ("default", "canonical"):
        raise ValueError("Unknown quadrature scheme: %s." % scheme)
    if scheme == "default":
        table = tabulated_rule(ref_el.get_spatial_dimension(), degree) \
            if ref_el.get_shape() in (reference_element.TRIANGLE, reference_element.TETRAHEDRON) else None
        if table is not None:
            return _TabulatedSimplexRule(ref_el, *table)
    return make_quadrature(ref_el, (degree + 2) // 2)
