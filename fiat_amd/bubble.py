"""Bubble elements (FIAT/bubble.py:13-44): the Lagrange dofs in the interior of the cell (Bubble) or of
its facets (FacetBubble), as a restriction of the Lagrange element."""
from itertools import chain

from .lagrange import Lagrange
from .restricted import RestrictedElement


class CodimBubble(RestrictedElement):
    def __init__(self, ref_el, degree, codim, variant=None, quad_scheme=None):
        if variant and variant.startswith("integral"):
            raise NotImplementedError("integral-variant bubbles build on IntegratedLegendre, out of scope for fiat_amd")
        element = Lagrange(ref_el, degree, variant=variant) if variant else Lagrange(ref_el, degree)
        cell_dim = ref_el.get_spatial_dimension()
        dofs = sorted(chain(*element.entity_dofs()[cell_dim - codim].values()))
        if len(dofs) == 0:
            raise RuntimeError('Bubble element of degree %d and codimension %d has no dofs' % (degree, codim))
        super().__init__(element, indices=dofs)


class Bubble(CodimBubble):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        super().__init__(ref_el, degree, codim=0, variant=variant, quad_scheme=quad_scheme)


class FacetBubble(CodimBubble):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        super().__init__(ref_el, degree, codim=1, variant=variant, quad_scheme=quad_scheme)
