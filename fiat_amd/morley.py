"""Morley elements on triangles and tetrahedra: P2 with the average over every sub-entity of codimension 2
(the vertex value on a triangle, the edge average on a tetrahedron) and the facet average of the normal
derivative, divided by (d-1)!.  Behaviour as FIAT/morley.py:17-66."""
import math

import numpy

from . import finite_element, functional, polynomial_set
from .dof_layout import DofLayout
from .quadrature import create_quadrature


def morley_dofs(cell, degree=2):
    lay = DofLayout(cell)
    sd = lay.sd
    if sd == 2:
        lay.lattice(0, degree, lambda _, pts: [functional.PointEvaluation(cell, x) for x in pts])
    else:
        lay.moments(sd - 2, 0, degree, lambda _: [numpy.float64(1.0)], tests=lambda rule: numpy.ones((1, len(rule.pts))))
    facet_rule = create_quadrature(cell.construct_subelement(sd - 1), degree - 1)
    weight = numpy.full(len(facet_rule.pts), 1.0 / math.factorial(sd - 1))
    for f in lay.entities(sd - 1):
        lay.place(sd - 1, f, [functional.IntegralMomentOfNormalDerivative(cell, f, facet_rule, weight)])
    return lay.dual_set()


class Morley(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=2):
        if ref_el.get_spatial_dimension() not in (2, 3):
            raise ValueError("Morley only defined on simplices of dimension >= 2")
        if degree != 2:
            raise ValueError(f"{type(self).__name__} only defined for degree == 2")
        super().__init__(polynomial_set.ONPolynomialSet(ref_el, degree), morley_dofs(ref_el, degree), degree)
