"""Raviart-Thomas H(div) element on simplices (FIAT/raviart_thomas.py:17-157):
space P_k^d + (homogeneous P_k) X; dofs = normal moments on facets and interior
moments ("integral" variant)."""
from itertools import chain

import numpy

from . import dual_set, expansions, finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule, create_quadrature


def RTSpace(ref_el, degree):
    sd = ref_el.get_spatial_dimension()
    k = degree - 1
    vec_Pkp1 = polynomial_set.ONPolynomialSet(ref_el, k + 1, (sd,))
    dimPkp1 = expansions.polynomial_dimension(ref_el, k + 1)
    dimPk = expansions.polynomial_dimension(ref_el, k)
    dimPkm1 = expansions.polynomial_dimension(ref_el, k - 1)
    vec_Pk = vec_Pkp1.take(list(chain(*(range(i * dimPkp1, i * dimPkp1 + dimPk) for i in range(sd)))))
    Pkp1 = polynomial_set.ONPolynomialSet(ref_el, k + 1)
    PkH = Pkp1.take(list(range(dimPkm1, dimPk)))
    Q = create_quadrature(ref_el, 2 * (k + 1))
    Qpts, Qwts = Q.get_points(), Q.get_weights()
    PkH_at_Qpts = PkH.tabulate(Qpts)[(0,) * sd]
    Pkp1_at_Qpts = Pkp1.tabulate(Qpts)[(0,) * sd]
    vals = PkH_at_Qpts[:, None, :] * Qpts.T[None, :, :]
    coeffs = numpy.dot(vals * Qwts, Pkp1_at_Qpts.T)
    PkHx = polynomial_set.PolynomialSet(ref_el, k, k + 1, vec_Pkp1.get_expansion_set(), coeffs)
    return polynomial_set.polynomial_set_union_normalized(vec_Pk, PkHx)


class RTDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, interpolant_deg, quad_scheme):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        nodes = []
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        if variant == "integral":
            facet = ref_el.construct_subelement(sd - 1)
            q = degree - 1
            Q_ref = parse_quadrature_scheme(facet, interpolant_deg + q, quad_scheme)
            Pq = polynomial_set.ONPolynomialSet(facet, q if sd > 1 else 0)
            Pq_at_qpts = Pq.tabulate(Q_ref.get_points())[(0,) * (sd - 1)]
            for f in sorted(top[sd - 1]):
                first = len(nodes)
                Q = FacetQuadratureRule(ref_el, sd - 1, f, Q_ref, avg=True)
                n = ref_el.compute_scaled_normal(f)
                phis = n[None, :, None] * Pq_at_qpts[:, None, :]
                nodes.extend(functional.FrobeniusIntegralMoment(ref_el, Q, phi) for phi in phis)
                entity_ids[sd - 1][f] = list(range(first, len(nodes)))
            if q > 0:
                cell = ref_el.construct_subelement(sd)
                Q_ref = parse_quadrature_scheme(cell, interpolant_deg + q - 1, quad_scheme)
                Pqm1_at_qpts = polynomial_set.ONPolynomialSet(cell, q - 1).tabulate(Q_ref.get_points())[(0,) * sd]
                for entity in sorted(top[sd]):
                    Q = FacetQuadratureRule(ref_el, sd, entity, Q_ref)
                    first = len(nodes)
                    nodes.extend(functional.IntegralMoment(ref_el, Q, phi, (d,), (sd,))
                                 for d in range(sd) for phi in Pqm1_at_qpts)
                    entity_ids[sd][entity] = list(range(first, len(nodes)))
        else:  # "point": scaled-normal evaluations on facet lattices, component evaluations on the interior lattice
            for f in sorted(top[sd - 1]):
                first = len(nodes)
                nodes.extend(functional.PointScaledNormalEvaluation(ref_el, f, pt)
                             for pt in ref_el.make_points(sd - 1, f, sd + degree - 1))
                entity_ids[sd - 1][f] = list(range(first, len(nodes)))
            if degree > 1:
                first = len(nodes)
                pts = ref_el.make_points(sd, 0, sd + degree - 1)
                nodes.extend(functional.ComponentPointEvaluation(ref_el, d, (sd,), pt) for d in range(sd) for pt in pts)
                entity_ids[sd][0] = list(range(first, len(nodes)))
        super().__init__(nodes, ref_el, entity_ids)


class RaviartThomas(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, interpolant_deg = check_format_variant(variant, degree)
        poly_set = RTSpace(ref_el, degree)
        dual = RTDualSet(ref_el, degree, variant, interpolant_deg, quad_scheme)
        super().__init__(poly_set, dual, degree, formdegree=ref_el.get_spatial_dimension() - 1,
                         mapping="contravariant piola")
