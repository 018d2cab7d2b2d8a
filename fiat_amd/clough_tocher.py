"""Hsieh-Clough-Tocher C1 macro element on the barycentric (Alfeld) split of a triangle.

Behaviour of FIAT/hct.py:19-88 -- same nodes in the same order, same prime space -- assembled from three
node groups:

* vertex jets: value and the two first derivatives at every vertex of the parent triangle;
* edge moments: averages over each edge of (normal derivative) x Jacobi(1,1) polynomials of degree <= k and of
  (function) x derivatives of those polynomials, k = degree - 3; the reduced element (degree 3) instead constrains
  the normal derivative against the quadratic Legendre polynomial;
* interior moments against P_(degree-4), integrated with the composite rule of the split.

Prime basis: ``macro.CkPolynomialSet`` (C1 across the spokes, C^(degree-1) at the barycentre) over the C0 macro
expansion set.  Riesz assembly -- point derivatives and derivative moments of the macro expansion set, with averaged
binning at the vertices, which lie on interfaces of the split --, Vandermonde solve and tabulation run on the device.
"""
import numpy

from . import dual_set, finite_element, functional, jacobi, macro, polynomial_set, quadrature
from .check_format_variant import parse_quadrature_scheme
from .reference_element import TRIANGLE, ufc_simplex


def _vertex_jets(cell, v):
    x = cell.get_vertices()[v]
    grads = [functional.PointDerivative(cell, x, alpha) for alpha in polynomial_set.mis(2, 1)]
    return [functional.PointEvaluation(cell, x)] + grads


def _edge_rule(degree, k, quad_scheme):
    """Rule on the reference edge and its abscissae mapped to (-1, 1)."""
    rule = parse_quadrature_scheme(ufc_simplex(1), degree - 1 + k, quad_scheme)
    return rule, 2.0 * rule.get_points() - 1.0


def _edge_moments(cell, e, rule, weights, dweights):
    Q = quadrature.FacetQuadratureRule(cell, 1, e, rule, avg=True)
    normal = cell.compute_normal(e)
    return ([functional.IntegralMomentOfDerivative(cell, Q, w, normal) for w in weights] +
            [functional.IntegralMoment(cell, Q, dw) for dw in dweights])


def _interior_moments(cell, split, degree, quad_scheme):
    q = degree - 4
    if q < 0:
        return []
    Q = parse_quadrature_scheme(split, degree + q, quad_scheme)
    tests = polynomial_set.ONPolynomialSet(cell, q, scale=1).tabulate(Q.get_points())[(0, 0)] / cell.volume()
    return [functional.IntegralMoment(cell, Q, f) for f in tests]


class HCTDualSet(dual_set.DualSet):
    def __init__(self, ref_complex, degree, reduced=False, quad_scheme=None):
        if reduced and degree != 3:
            raise ValueError("Reduced HCT only defined for degree = 3")
        if degree < 3:
            raise ValueError("HCT only defined for degree >= 3")
        cell = ref_complex.get_parent()
        if cell.get_shape() != TRIANGLE:
            raise ValueError("HCT only defined on triangles")
        top = cell.get_topology()
        groups = [((0, v), _vertex_jets(cell, v)) for v in sorted(top[0])]
        if reduced:
            rule, x = _edge_rule(degree, 2, quad_scheme)
            legendre2 = jacobi.eval_jacobi_batch(0, 0, 2, x)[2]
            groups += [((1, e), [functional.IntegralMomentOfNormalDerivative(cell, e, rule, legendre2)])
                       for e in sorted(top[1])]
        else:
            k = degree - 3
            rule, x = _edge_rule(degree, k, quad_scheme)
            weights = jacobi.eval_jacobi_batch(1, 1, k, x)
            dweights = 2 * jacobi.eval_jacobi_deriv_batch(1, 1, k, x)[1:]
            groups += [((1, e), _edge_moments(cell, e, rule, weights, dweights)) for e in sorted(top[1])]
            groups.append(((2, 0), _interior_moments(cell, ref_complex, degree, quad_scheme)))
        nodes = []
        entity_ids = {dim: {entity: [] for entity in sorted(top[dim])} for dim in sorted(top)}
        for (dim, entity), new in groups:
            entity_ids[dim][entity] = list(range(len(nodes), len(nodes) + len(new)))
            nodes.extend(new)
        super().__init__(nodes, cell, entity_ids)


class HsiehCloughTocher(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=3, reduced=False, quad_scheme=None):
        split = macro.AlfeldSplit(ref_el)
        prime = macro.CkPolynomialSet(split, degree, order=1, vorder=degree - 1, variant="bubble")
        super().__init__(prime, HCTDualSet(split, degree, reduced=reduced, quad_scheme=quad_scheme), degree, formdegree=0)
