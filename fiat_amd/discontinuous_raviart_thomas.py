"""Discontinuous Raviart-Thomas (FIAT/discontinuous_raviart_thomas.py:11-62): the RT space with
scaled-normal point evaluations on the facet lattices and component evaluations on the interior
lattice, every dof associated with the cell."""
from . import dual_set, finite_element, functional
from .raviart_thomas import RTSpace


class DRTDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        nodes = []
        for f in sorted(top[sd - 1]):
            nodes.extend(functional.PointScaledNormalEvaluation(ref_el, f, pt)
                         for pt in ref_el.make_points(sd - 1, f, sd + degree - 1))
        if degree > 1:
            pts = ref_el.make_points(sd, 0, sd + degree - 1)
            nodes.extend(functional.ComponentPointEvaluation(ref_el, d, (sd,), pt) for d in range(sd) for pt in pts)
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        entity_ids[sd][0] = list(range(len(nodes)))
        super().__init__(nodes, ref_el, entity_ids)


class DiscontinuousRaviartThomas(finite_element.CiarletElement):
    def __init__(self, ref_el, degree):
        super().__init__(RTSpace(ref_el, degree), DRTDualSet(ref_el, degree), degree, mapping="contravariant piola")
