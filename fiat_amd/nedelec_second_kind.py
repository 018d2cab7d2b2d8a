"""Nedelec second-kind H(curl) element on triangles and tetrahedra
(FIAT/nedelec_second_kind.py:20-222): the full space P_k^d; dofs on every entity of dimension
m = 1 .. d are Frobenius moments against the contravariantly mapped vector polynomials of degree
k - m + 1 of that entity: all of P_k on edges, Raviart-Thomas RT_{k-m+1} on faces and in the cell
(:107-158); "point" variant: tangential point evaluations on the edges instead of the edge moments.  Tabulation runs on the same kernels as every other
coeffs x Dubiner element."""
import numpy

from . import dual_set, finite_element, functional, polynomial_set, raviart_thomas
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule


def _entity_test_functions(entity_cell, dim, deg, variant, qpts):
    """Vector test functions on the reference entity at ``qpts``: (nfun, dim, nq)."""
    if dim == 1:
        space = polynomial_set.ONPolynomialSet(entity_cell, deg, (dim,))
    else:
        space = raviart_thomas.RaviartThomas(entity_cell, deg, variant).get_nodal_basis()
    return space.tabulate(qpts)[(0,) * dim]


class NedelecSecondKindDual(dual_set.DualSet):
    def __init__(self, cell, degree, variant, interpolant_deg, quad_scheme):
        d = cell.get_spatial_dimension()
        assert d in (2, 3), "Second kind Nedelecs only implemented in 2/3D."
        top = cell.get_topology()
        ids = {dim: {entity: [] for entity in sorted(top[dim])} for dim in top}
        dofs = []
        if interpolant_deg is None:
            interpolant_deg = degree
        if variant == "point":  # tangential evaluations at degree + 1 points of every edge (:96-110)
            for edge in sorted(top[1]):
                points = cell.make_points(1, edge, degree + 2)
                ids[1][edge] = list(range(len(dofs), len(dofs) + len(points)))
                dofs += [functional.PointEdgeTangentEvaluation(cell, edge, pt) for pt in points]
        for dim in range(2 if variant == "point" else 1, d + 1):
            test_degree = degree - dim + 1
            if test_degree < 1:
                continue
            entity_cell = cell.construct_subelement(dim)
            Q_ref = parse_quadrature_scheme(entity_cell, interpolant_deg + test_degree, quad_scheme)
            Phis = _entity_test_functions(entity_cell, dim, test_degree, variant, Q_ref.get_points())
            for entity in sorted(top[dim]):
                Q = FacetQuadratureRule(cell, dim, entity, Q_ref)
                piola = Q.jacobian() / Q.jacobian_determinant()  # reference entity -> entity of the cell
                mapped = numpy.einsum("ab,ibq->iaq", piola, Phis)
                ids[dim][entity] = list(range(len(dofs), len(dofs) + len(mapped)))
                dofs += [functional.FrobeniusIntegralMoment(cell, Q, phi) for phi in mapped]
        super().__init__(dofs, cell, ids)


class NedelecSecondKind(finite_element.CiarletElement):
    """N2curl_k, k >= 1; variant in {None, "integral", "integral(q)"}."""

    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, interpolant_deg = check_format_variant(variant, degree)
        assert degree >= 1, "Second kind Nedelecs start at 1!"
        d = ref_el.get_spatial_dimension()
        poly_set = polynomial_set.ONPolynomialSet(ref_el, degree, (d,))
        dual = NedelecSecondKindDual(ref_el, degree, variant, interpolant_deg, quad_scheme)
        super().__init__(poly_set, dual, degree, formdegree=1, mapping="covariant piola")
