"""Crouzeix-Raviart element (FIAT/crouzeix_raviart.py:18-94): P_k, k odd; "integral" variants: moments
against orthonormal P_{k-1} on every facet and P_{k-1-dim} on the higher-dimensional entities
(k > 1 on triangles only)."""
from . import dual_set, finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule


class CrouzeixRaviartDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, interpolant_deg, quad_scheme):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        if sd < 2:
            raise NotImplementedError("Crouzeix-Raviart needs a triangle or a tetrahedron in fiat_amd")
        if degree > 1 and sd != 2:
            raise NotImplementedError("High-order Crouzeix-Raviart is only implemented on triangles.")
        if variant != "integral":
            raise NotImplementedError("CrouzeixRaviart: only the 'integral' variants are supported by fiat_amd")
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        nodes = []
        for dim in range(1, sd + 1):
            k = degree - 1 if dim == sd - 1 else degree - (1 + dim)
            if k < 0:
                continue
            entity_cell = ref_el.construct_subelement(dim)
            Q_ref = parse_quadrature_scheme(entity_cell, k + interpolant_deg, quad_scheme)
            phis = polynomial_set.ONPolynomialSet(entity_cell, k).tabulate(Q_ref.get_points())[(0,) * dim]
            for i in sorted(top[dim]):
                Q = FacetQuadratureRule(ref_el, dim, i, Q_ref, avg=True)
                entity_ids[dim][i] = list(range(len(nodes), len(nodes) + len(phis)))
                nodes += [functional.IntegralMoment(ref_el, Q, phi) for phi in phis]
        super().__init__(nodes, ref_el, entity_ids)


class CrouzeixRaviart(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        if degree % 2 != 1:
            raise ValueError("Crouzeix-Raviart only defined for odd degree")
        _, variant, interpolant_deg = check_format_variant(variant, degree)
        poly_set = polynomial_set.ONPolynomialSet(ref_el, degree)
        dual = CrouzeixRaviartDualSet(ref_el, degree, variant, interpolant_deg, quad_scheme)
        super().__init__(poly_set, dual, degree)
