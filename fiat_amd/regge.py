"""Generalised Regge elements (FIAT/regge.py:11-79): symmetric-matrix-valued P_k with tangential-tangential
continuity.  On every sub-entity of dimension m >= 1 and for each of its edge tangents t: moments of
t^T u t against orthonormal P_{k-m+1} of the entity ("integral"), or its values on a lattice ("point").
Tabulation: rows = ndof * sd * sd on the same kernels as every other coeffs x Dubiner element."""
from . import dual_set, finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule


class ReggeDual(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, qdegree, quad_scheme):
        top = ref_el.get_topology()
        entity_ids = {dim: {i: [] for i in sorted(top[dim])} for dim in sorted(top)}
        nodes = []
        for dim in sorted(top):
            if dim == 0:
                continue
            if variant == "point":
                per_entity = lambda entity, tangents: [functional.PointwiseInnerProductEvaluation(ref_el, t, t, pt)
                                                       for pt in ref_el.make_points(dim, entity, degree + 2) for t in tangents]
            else:
                k = degree - dim + 1
                if k < 0:
                    continue
                facet = ref_el.construct_subelement(dim)
                Q = parse_quadrature_scheme(facet, qdegree + k, quad_scheme)
                phis = polynomial_set.ONPolynomialSet(facet, k).tabulate(Q.get_points())[(0,) * dim]

                def per_entity(entity, tangents, Q=Q, phis=phis, dim=dim):
                    Q_mapped = FacetQuadratureRule(ref_el, dim, entity, Q, avg=True)
                    return [functional.TensorBidirectionalIntegralMoment(ref_el, t, t, Q_mapped, phi) for phi in phis for t in tangents]
            for entity in sorted(top[dim]):
                new = per_entity(entity, ref_el.compute_face_edge_tangents(dim, entity))
                entity_ids[dim][entity] = list(range(len(nodes), len(nodes) + len(new)))
                nodes += new
        super().__init__(nodes, ref_el, entity_ids)


class Regge(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=0, variant=None, quad_scheme=None):
        if degree < 0:
            raise ValueError(f"{type(self).__name__} only defined for degree >= 0")
        _, variant, qdegree = check_format_variant(variant, degree)
        poly_set = polynomial_set.ONSymTensorPolynomialSet(ref_el, degree)
        dual = ReggeDual(ref_el, degree, variant, qdegree, quad_scheme)
        super().__init__(poly_set, dual, degree, formdegree=(1, 1), mapping="double covariant piola")
