"""Brezzi-Douglas-Fortin-Marini (FIAT/brezzi_douglas_fortin_marini.py:7-25): BDM_k restricted to the
facet dofs that test against P_{k-1} (the first dim P_{k-1}(facet) normal moments of each facet;
the orthonormal facet basis is ordered by degree) plus all interior dofs."""
from .brezzi_douglas_marini import BrezziDouglasMarini
from .expansions import polynomial_dimension
from .restricted import RestrictedElement


def BrezziDouglasFortinMarini(ref_el, degree, variant=None, quad_scheme=None):
    if variant == "point":  # interior dofs of BDM_k + facet dofs of BDM_{k-1} (:12-15)
        from .nodal_enriched import NodalEnrichedElement
        interior = RestrictedElement(BrezziDouglasMarini(ref_el, degree, variant=variant), restriction_domain="interior")
        facets = RestrictedElement(BrezziDouglasMarini(ref_el, degree - 1, variant=variant), restriction_domain="facet")
        return NodalEnrichedElement(interior, facets)
    bdm = BrezziDouglasMarini(ref_el, degree, variant=variant, quad_scheme=quad_scheme)
    entity_ids = bdm.get_dual_set().get_entity_ids()
    sd = ref_el.get_spatial_dimension()
    indices = []
    for dim in sorted(entity_ids):
        keep = slice(polynomial_dimension(ref_el.construct_subelement(dim), degree - 1)) if dim == sd - 1 else slice(None)
        for entity in sorted(entity_ids[dim]):
            indices.extend(entity_ids[dim][entity][keep])
    return RestrictedElement(bdm, indices)
