"""Brezzi-Douglas-Marini H(div) element on triangles and tetrahedra
(FIAT/brezzi_douglas_marini.py:15-112): the full space P_k^d; dofs = normal moments against
P_k on every facet (:30-43) and, for k > 1, interior moments against the first-kind Nedelec
functions of degree k - 1 (:56-71); "point" variant: scaled-normal point evaluations on the
facets (:45-54).  Tabulation runs on the same
kernels as every other coeffs x Dubiner element."""
import numpy

from . import dual_set, finite_element, functional, nedelec, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule


def _normal_moments(ref_el, degree, quad_degree, quad_scheme):
    """[(facet, nodes)]: f -> average over the facet of (f . n) p for p in an orthonormal basis of P_degree."""
    sd = ref_el.get_spatial_dimension()
    facet_cell = ref_el.construct_subelement(sd - 1)
    Q_ref = parse_quadrature_scheme(facet_cell, quad_degree, quad_scheme)
    p_at_q = polynomial_set.ONPolynomialSet(facet_cell, degree).tabulate(Q_ref.get_points())[(0,) * (sd - 1)]
    out = []
    for f in sorted(ref_el.get_topology()[sd - 1]):
        Q = FacetQuadratureRule(ref_el, sd - 1, f, Q_ref, avg=True)
        normal = numpy.asarray(ref_el.compute_scaled_normal(f), dtype=float)
        out.append((f, [functional.FrobeniusIntegralMoment(ref_el, Q, numpy.outer(normal, p)) for p in p_at_q]))
    return out


class BDMDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, interpolant_deg, quad_scheme):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        nodes = []
        if variant == "integral":
            facet_nodes = _normal_moments(ref_el, degree, interpolant_deg + degree, quad_scheme)
        else:  # "point": scaled-normal evaluations on the facet lattices
            facet_nodes = [(f, [functional.PointScaledNormalEvaluation(ref_el, f, pt)
                                for pt in ref_el.make_points(sd - 1, f, sd + degree)]) for f in sorted(top[sd - 1])]
        for f, moments in facet_nodes:
            entity_ids[sd - 1][f] = list(range(len(nodes), len(nodes) + len(moments)))
            nodes += moments
        if degree > 1:
            if interpolant_deg is None:
                interpolant_deg = degree
            cell = ref_el.construct_subelement(sd)
            Q_ref = parse_quadrature_scheme(cell, interpolant_deg + degree - 1, quad_scheme)
            ned = nedelec.Nedelec(cell, degree - 1, variant).tabulate(0, Q_ref.get_points())[(0,) * sd]
            for entity in sorted(top[sd]):
                Q = FacetQuadratureRule(ref_el, sd, entity, Q_ref)
                # test functions pulled to the cell covariantly: J^{-T} N_i
                mapped = numpy.einsum("ba,ibq->iaq", numpy.linalg.inv(Q.jacobian()), ned)
                entity_ids[sd][entity] = list(range(len(nodes), len(nodes) + len(mapped)))
                nodes += [functional.FrobeniusIntegralMoment(ref_el, Q, phi) for phi in mapped]
        super().__init__(nodes, ref_el, entity_ids)


class BrezziDouglasMarini(finite_element.CiarletElement):
    """BDM_k, k >= 1; variant in {None, "integral", "integral(q)"}."""

    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, interpolant_deg = check_format_variant(variant, degree)
        if degree < 1:
            raise Exception("BDM_k elements only valid for k >= 1")
        sd = ref_el.get_spatial_dimension()
        if sd not in (2, 3):
            raise NotImplementedError("BrezziDouglasMarini needs a triangle or a tetrahedron")
        poly_set = polynomial_set.ONPolynomialSet(ref_el, degree, (sd,))
        dual = BDMDualSet(ref_el, degree, variant, interpolant_deg, quad_scheme)
        super().__init__(poly_set, dual, degree, formdegree=sd - 1, mapping="contravariant piola")
