"""Direct sum of nodal elements, re-orthogonalised against the merged dual basis
(FIAT/nodal_enriched.py:20-146): the primal coefficients of the summands are placed into the
coefficient array of the richest expansion set (hierarchical numbering: the members of a lower
degree are a prefix per entity, `polynomial_entity_ids`), the dual bases are concatenated, and the
CiarletElement constructor does the rest (Vandermonde solve on the device)."""
import math

import numpy

from .dual_set import DualSet
from .expansions import polynomial_entity_ids
from .finite_element import CiarletElement
from .polynomial_set import PolynomialSet


def _merge_coeffs(coeffss, ref_el, degrees, continuity):
    sd = ref_el.get_spatial_dimension()
    entity_ids = polynomial_entity_ids(ref_el, max(degrees), continuity)
    value_shape = coeffss[0].shape[1:-1]
    assert all(c.shape[1:-1] == value_shape for c in coeffss)
    merged = numpy.zeros((sum(c.shape[0] for c in coeffss), *value_shape, max(c.shape[-1] for c in coeffss)))
    row = 0
    for c, degree in zip(coeffss, degrees):
        members = []
        for dim in (sorted(entity_ids) if continuity == "C0" else (sd,)):
            count = math.comb(degree - 1, dim) if continuity == "C0" else math.comb(degree + dim, dim)
            for entity in sorted(entity_ids[dim]):
                members.extend(entity_ids[dim][entity][:count])
        merged[row:row + c.shape[0], ..., members] = c
        row += c.shape[0]
    return merged


def _merge_entity_ids(all_ids, offsets):
    merged = {}
    for ids, offset in zip(all_ids, offsets):
        for dim, entities in ids.items():
            for entity, dofs in entities.items():
                merged.setdefault(dim, {}).setdefault(entity, []).extend(int(offset) + d for d in dofs)
    return merged


class NodalEnrichedElement(CiarletElement):
    def __init__(self, *elements):
        if not all(e.is_nodal() for e in elements):
            raise ValueError("Not all elements given for construction of NodalEnrichedElement are nodal")
        degrees = [e.degree() for e in elements]
        embedded_degree = max(degrees)
        order = max(e.get_order() for e in elements)
        formdegree = None if any(e.get_formdegree() is None for e in elements) else max(e.get_formdegree() for e in elements)
        richest = max(elements, key=lambda e: e.degree())
        ref_el = richest.get_reference_element()
        expansion_set = richest.get_nodal_basis().get_expansion_set()
        mapping = richest.mapping()[0]
        value_shape = richest.value_shape()
        assert all(set(e.mapping()) == {mapping} for e in elements)
        assert all(e.value_shape() == value_shape for e in elements)
        if not all(e.get_nodal_basis().get_expansion_set() == expansion_set for e in elements):
            raise NotImplementedError("NodalEnrichedElement of elements over different expansion sets (projection route) "
                                      "is out of scope for fiat_amd")
        coeffs = _merge_coeffs([e.get_coeffs() for e in elements], ref_el, degrees, expansion_set.continuity)
        poly_set = PolynomialSet(ref_el, embedded_degree, embedded_degree, expansion_set, coeffs)
        offsets = numpy.cumsum([0] + [e.space_dimension() for e in elements[:-1]])
        entity_ids = _merge_entity_ids((e.entity_dofs() for e in elements), offsets)
        nodes = [node for e in elements for node in e.dual_basis()]
        super().__init__(poly_set, DualSet(nodes, ref_el, entity_ids), order, formdegree=formdegree, mapping=mapping)
