"""Discontinuous Lagrange element on simplices (FIAT/discontinuous_lagrange.py
:147-241): the CG lattice nodes, all owned by the cell; prime basis = orthonormal
Dubiner set.  Degree 0 is the P0 element (FIAT/P0.py:17-52)."""
import numpy

from . import dual_set, finite_element, functional, polynomial_set
from .barycentric_interpolation import LagrangePolynomialSet, get_lagrange_points
from .check_format_variant import parse_lagrange_variant
from .reference_element import LINE, make_lattice


class BrokenLagrangeDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, point_variant="equispaced"):
        top = ref_el.get_topology()
        nodes = []
        entity_ids = {}
        for dim in sorted(top):
            entity_ids[dim] = {}
            for entity in sorted(top[dim]):
                pts = ref_el.make_points(dim, entity, degree, variant=point_variant)
                nodes.extend(functional.PointEvaluation(ref_el, x) for x in pts)
                entity_ids[dim][entity] = []
        entity_ids[max(top)][0] = list(range(len(nodes)))
        super().__init__(nodes, ref_el, entity_ids)


class DiscontinuousLagrangeDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, point_variant="equispaced"):
        top = ref_el.get_topology()
        sd = ref_el.get_dimension()
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        nodes = []
        for cell in sorted(top[sd]):    # every cell of a complex owns its own lattice
            first = len(nodes)
            pts = make_lattice(ref_el.get_vertices_of_subcomplex(top[sd][cell]), degree, variant=point_variant)
            nodes.extend(functional.PointEvaluation(ref_el, x) for x in pts)
            entity_ids[sd][cell] = list(range(first, len(nodes)))
        super().__init__(nodes, ref_el, entity_ids)


class P0Dual(dual_set.DualSet):
    def __init__(self, ref_el):
        top = ref_el.get_topology()
        sd = ref_el.get_spatial_dimension()
        bary = tuple(numpy.average(numpy.asarray(ref_el.get_vertices()), axis=0))
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        entity_ids[sd][0] = [0]
        super().__init__([functional.PointEvaluation(ref_el, bary)], ref_el, entity_ids)


class P0(finite_element.CiarletElement):
    def __init__(self, ref_el):
        poly_set = polynomial_set.ONPolynomialSet(ref_el, 0)
        super().__init__(poly_set, P0Dual(ref_el), 0, formdegree=ref_el.get_spatial_dimension())


class DiscontinuousLagrange(finite_element.CiarletElement):
    def __new__(cls, ref_el, degree, variant="equispaced"):
        if degree == 0:
            splitting, _ = parse_lagrange_variant(variant, discontinuous=True)
            if splitting is None and not ref_el.is_macrocell():
                return P0(ref_el)
        return super().__new__(cls)

    def __init__(self, ref_el, degree, variant="equispaced"):
        splitting, point_variant = parse_lagrange_variant(variant, discontinuous=True)
        if splitting is not None:
            ref_el = splitting(ref_el)
        if ref_el.is_macrocell() and ref_el.get_shape() == LINE:
            raise NotImplementedError("macro Lagrange elements on intervals")
        if point_variant in ("equispaced", "gll", "lgc"):
            dual = BrokenLagrangeDualSet(ref_el, degree, point_variant=point_variant)
        else:
            dual = DiscontinuousLagrangeDualSet(ref_el, degree, point_variant=point_variant)
        if ref_el.get_shape() == LINE:
            poly_set = LagrangePolynomialSet(ref_el, get_lagrange_points(dual))
        else:
            poly_set = polynomial_set.ONPolynomialSet(ref_el, degree)
        super().__init__(poly_set, dual, degree, formdegree=ref_el.get_spatial_dimension())


class GaussLegendre(DiscontinuousLagrange):
    """Discontinuous element with nodes at the (recursive) Gauss-Legendre points (FIAT/gauss_legendre.py)."""

    def __init__(self, ref_el, degree):
        DiscontinuousLagrange.__init__(self, ref_el, degree, variant="gl")
