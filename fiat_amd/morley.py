"""Morley element on triangles and tetrahedra (FIAT/morley.py:17-66): P2; dofs = average over every
codimension-2 entity (vertex values on a triangle, edge averages on a tetrahedron) and the
average normal derivative over every facet (scaled by 1/(sd-1)!, :47-48)."""
import math

import numpy

from . import dual_set, finite_element, functional, polynomial_set
from .quadrature import FacetQuadratureRule, create_quadrature


class MorleyDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        nodes = []
        # codimension 2: integral average (a point evaluation when the entity is a vertex)
        dim = sd - 2
        if dim > 0:
            Q_ref = create_quadrature(ref_el.construct_subelement(dim), degree)
            one = numpy.ones(len(Q_ref.get_weights()))
        for entity in sorted(top[dim]):
            if dim == 0:
                node = functional.PointEvaluation(ref_el, ref_el.get_vertices()[top[0][entity][0]])
            else:
                node = functional.IntegralMoment(ref_el, FacetQuadratureRule(ref_el, dim, entity, Q_ref, avg=True), one)
            entity_ids[dim][entity] = [len(nodes)]
            nodes.append(node)
        # codimension 1: average of the normal derivative
        Q_ref = create_quadrature(ref_el.construct_subelement(sd - 1), degree - 1)
        scale = numpy.ones(len(Q_ref.get_weights())) / math.factorial(sd - 1)
        for entity in sorted(top[sd - 1]):
            entity_ids[sd - 1][entity] = [len(nodes)]
            nodes.append(functional.IntegralMomentOfNormalDerivative(ref_el, entity, Q_ref, scale))
        super().__init__(nodes, ref_el, entity_ids)


class Morley(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=2):
        if ref_el.get_spatial_dimension() not in (2, 3):
            raise ValueError("Morley only defined on simplices of dimension >= 2")
        if degree != 2:
            raise ValueError(f"{type(self).__name__} only defined for degree == 2")
        poly_set = polynomial_set.ONPolynomialSet(ref_el, degree)
        super().__init__(poly_set, MorleyDualSet(ref_el, degree), degree)
