"""Argyris element on triangles (FIAT/argyris.py:14-114): P_k, k >= 5, bubble-variant expansion set; dofs = the
second-order jet at every vertex (derivative functionals of order 1 and 2: rows of the Vandermonde matrix from the
order-2 tables of the expansion set, dual_set.to_riesz), then per edge, "integral": moments of the normal derivative
against Jacobi(2,2) polynomials and of the function against their derivatives, interior moments against P_{k-6};
"point": normal derivatives and values on edge lattices, values on an interior lattice."""
from . import dual_set, finite_element, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .functional import IntegralMoment, IntegralMomentOfDerivative, PointDerivative, PointEvaluation, PointNormalDerivative
from .jacobi import eval_jacobi_batch, eval_jacobi_deriv_batch
from .quadrature import FacetQuadratureRule
from .reference_element import ufc_simplex


class ArgyrisDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, interpolant_deg, quad_scheme):
        if ref_el.get_spatial_dimension() != 2:
            raise ValueError("Argyris only defined on triangles")
        top = ref_el.get_topology()
        entity_ids = {dim: {entity: [] for entity in sorted(top[dim])} for dim in sorted(top)}
        nodes = []

        def add(dim, entity, new):
            entity_ids[dim][entity] = entity_ids[dim][entity] + list(range(len(nodes), len(nodes) + len(new)))
            nodes.extend(new)

        verts = ref_el.get_vertices()
        for v in sorted(top[0]):
            add(0, v, [PointEvaluation(ref_el, verts[v])] +
                [PointDerivative(ref_el, verts[v], alpha) for alpha in ((1, 0), (0, 1), (2, 0), (1, 1), (0, 2))])
        if variant == "integral":
            k = degree - 5
            rline = ufc_simplex(1)
            Q_ref = parse_quadrature_scheme(rline, interpolant_deg + k - 1, quad_scheme)
            xref = 2.0 * Q_ref.get_points() - 1.0            # edge coordinate in (-1, 1)
            phis = eval_jacobi_batch(2, 2, k, xref)
            dphis = 2 * eval_jacobi_deriv_batch(2, 2, k, xref)
            for e in sorted(top[1]):
                Q = FacetQuadratureRule(ref_el, 1, e, Q_ref, avg=True)
                n = ref_el.compute_normal(e)
                add(1, e, [IntegralMomentOfDerivative(ref_el, Q, phi, n) for phi in phis] +
                    [IntegralMoment(ref_el, Q, dphi) for dphi in dphis[1:]])
            q = degree - 6
            if q >= 0:
                cell = ref_el.construct_subelement(2)
                Q_ref = parse_quadrature_scheme(cell, interpolant_deg + q, quad_scheme)
                phis = polynomial_set.ONPolynomialSet(cell, q, scale=1).tabulate(Q_ref.get_points())[(0, 0)]
                for entity in sorted(top[2]):
                    Q = FacetQuadratureRule(ref_el, 2, entity, Q_ref, avg=True)
                    add(2, entity, [IntegralMoment(ref_el, Q, phi) for phi in phis])
        elif variant == "point":
            for e in sorted(top[1]):
                add(1, e, [PointNormalDerivative(ref_el, e, pt) for pt in ref_el.make_points(1, e, degree - 3)] +
                    [PointEvaluation(ref_el, pt) for pt in ref_el.make_points(1, e, degree - 4)])
            if degree > 5:
                for entity in sorted(top[2]):
                    add(2, entity, [PointEvaluation(ref_el, pt) for pt in ref_el.make_points(2, entity, degree - 3)])
        else:
            raise ValueError("Invalid variant for Argyris")
        super().__init__(nodes, ref_el, entity_ids)


class Argyris(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=5, variant=None, quad_scheme=None):
        _, variant, interpolant_deg = check_format_variant(variant, degree)
        poly_set = polynomial_set.ONPolynomialSet(ref_el, degree, variant="bubble")
        dual = ArgyrisDualSet(ref_el, degree, variant, interpolant_deg, quad_scheme)
        super().__init__(poly_set, dual, degree)
