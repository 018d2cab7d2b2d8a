"""Bubble elements (FIAT/bubble.py:13-44): those Lagrange basis functions whose nodes lie in the interior of
the cell (`Bubble`) or in the interior of its facets (`FacetBubble`), obtained by restricting the Lagrange
element of the same degree to those degrees of freedom."""
from .lagrange import Lagrange
from .restricted import RestrictedElement


def _interior_dofs(element, dim):
    """Degrees of freedom attached to the entities of dimension `dim` themselves, in entity order."""
    per_entity = element.entity_dofs()[dim]
    return [dof for entity in sorted(per_entity) for dof in per_entity[entity]]


class CodimBubble(RestrictedElement):
    """Lagrange functions supported in the interior of the entities of a given codimension."""

    def __init__(self, ref_el, degree, codim, variant=None, quad_scheme=None):
        if variant is not None and variant.startswith("integral"):
            raise NotImplementedError("integral-variant bubbles build on IntegratedLegendre, out of scope for fiat_amd")
        lagrange = Lagrange(ref_el, degree) if variant is None else Lagrange(ref_el, degree, variant=variant)
        keep = sorted(_interior_dofs(lagrange, ref_el.get_spatial_dimension() - codim))
        if not keep:
            raise RuntimeError('Bubble element of degree %d and codimension %d has no dofs' % (degree, codim))
        RestrictedElement.__init__(self, lagrange, indices=keep)


class Bubble(CodimBubble):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        CodimBubble.__init__(self, ref_el, degree, 0, variant=variant, quad_scheme=quad_scheme)


class FacetBubble(CodimBubble):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        CodimBubble.__init__(self, ref_el, degree, 1, variant=variant, quad_scheme=quad_scheme)
