"""Hellan-Herrmann-Johnson elements (FIAT/hellan_herrmann_johnson.py:11-120): symmetric-matrix-valued P_k with
normal-normal continuity.  "integral": moments of n_f^T u n_f against orthonormal P_k on every facet f;
interior: n_f^T u n_f against P_{k-1} for every facet normal and n_i^T u n_j (i != j) against P_k.
"point" (tetrahedra): the same bilinear forms on lattices."""
from . import dual_set, finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule


class HellanHerrmannJohnsonDual(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, qdegree, quad_scheme):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        entity_ids = {dim: {i: [] for i in sorted(top[dim])} for dim in sorted(top)}
        nodes = []
        facets = sorted(top[sd - 1])
        n = [ref_el.compute_scaled_normal(f) for f in facets]
        mixed = [(facets[i + 1], facets[i + 2]) for i in range((sd - 1) * (sd - 2))]  # off-diagonal normal pairs (3-D)

        def add(dim, entity, new):
            entity_ids[dim][entity] = entity_ids[dim][entity] + list(range(len(nodes), len(nodes) + len(new)))
            nodes.extend(new)

        if variant == "point":
            if sd == 2:
                raise NotImplementedError("the 2-D point variant of HHJ keeps Cartesian interior dofs whose Riesz rows "
                                          "rely on NumPy fancy indexing in the reference; use the integral variant")
            ev = functional.PointwiseInnerProductEvaluation
            for f in facets:
                add(sd - 1, f, [ev(ref_el, n[f], n[f], pt) for pt in ref_el.make_points(sd - 1, f, degree + sd)])
            for entity in sorted(top[sd]):
                add(sd, entity, [ev(ref_el, n[f], n[f], pt) for pt in ref_el.make_points(sd, entity, degree + sd) for f in facets])
                add(sd, entity, [ev(ref_el, n[a], n[b], pt) for pt in ref_el.make_points(sd, entity, degree + sd + 1)
                                 for a, b in mixed])
        else:
            moment = functional.TensorBidirectionalIntegralMoment
            facet_cell = ref_el.construct_subelement(sd - 1)
            Q_ref = parse_quadrature_scheme(facet_cell, qdegree + degree, quad_scheme)
            Phis = polynomial_set.ONPolynomialSet(facet_cell, degree).tabulate(Q_ref.get_points())[(0,) * (sd - 1)]
            for f in facets:
                Q = FacetQuadratureRule(ref_el, sd - 1, f, Q_ref, avg=True)
                add(sd - 1, f, [moment(ref_el, n[f], n[f], Q, phi) for phi in Phis])
            cell = ref_el.construct_subelement(sd)
            Q_ref = parse_quadrature_scheme(cell, qdegree + degree, quad_scheme)
            P = polynomial_set.ONPolynomialSet(cell, degree)
            Phis = P.tabulate(Q_ref.get_points())[(0,) * sd]
            dimPkm1 = P.get_expansion_set().get_num_members(degree - 1) if degree >= 1 else 0
            for entity in sorted(top[sd]):
                Q = FacetQuadratureRule(ref_el, sd, entity, Q_ref, avg=True)
                add(sd, entity, [moment(ref_el, n[f], n[f], Q, phi) for phi in Phis[:dimPkm1] for f in facets])
                add(sd, entity, [moment(ref_el, n[a], n[b], Q, phi) for phi in Phis for a, b in mixed])
        super().__init__(nodes, ref_el, entity_ids)


class HellanHerrmannJohnson(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=0, variant=None, quad_scheme=None):
        if degree < 0:
            raise ValueError(f"{type(self).__name__} only defined for degree >= 0")
        _, variant, qdegree = check_format_variant(variant, degree)
        poly_set = polynomial_set.ONSymTensorPolynomialSet(ref_el, degree)
        dual = HellanHerrmannJohnsonDual(ref_el, degree, variant, qdegree, quad_scheme)
        sd = ref_el.get_spatial_dimension()
        super().__init__(poly_set, dual, degree, formdegree=(sd - 1, sd - 1), mapping="double contravariant piola")
