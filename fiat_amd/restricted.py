"""An element restricted to a subset of its degrees of freedom (FIAT/restricted.py:13-105): the
nodal basis functions of the kept dofs span the space, the kept functionals are the dual set.
Building block of the bubble elements and of Brezzi-Douglas-Fortin-Marini."""
from .dual_set import DualSet
from .finite_element import CiarletElement


class RestrictedDualSet(DualSet):
    def __init__(self, dual, indices):
        keep = sorted(indices)
        position = {dof: i for i, dof in enumerate(keep)}
        entity_ids = {d: {entity: [position[dof] for dof in dofs if dof in position] for entity, dofs in entities.items()}
                      for d, entities in dual.get_entity_ids().items()}
        self._dual = dual
        old_nodes = dual.get_nodes()
        super().__init__([old_nodes[i] for i in keep], dual.get_reference_element(), entity_ids)


class RestrictedElement(CiarletElement):
    def __init__(self, element, indices=None, restriction_domain=None, take_closure=True):
        if not (indices or restriction_domain):
            raise RuntimeError("Either indices or restriction_domain must be passed in")
        if not indices:
            indices = element.get_dual_set().get_indices(restriction_domain, take_closure=take_closure)
        if isinstance(indices, str):
            raise RuntimeError("variable 'indices' was a string; did you forget to use a keyword?")
        if len(indices) == 0:
            raise ValueError("No point in creating empty RestrictedElement.")
        self._element = element
        self._indices = indices
        poly_set = element.get_nodal_basis().take(indices)
        dual = RestrictedDualSet(element.get_dual_set(), indices)
        mappings = [element.mapping()[dof] for dof in indices]
        assert all(m == mappings[0] for m in mappings)
        super().__init__(poly_set, dual, element.degree(), element.get_formdegree(), mappings[0])
