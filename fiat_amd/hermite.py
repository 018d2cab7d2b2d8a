"""Cubic Hermite element on simplices (FIAT/hermite.py:11-80): P3; dofs = value and gradient at every
vertex, value at the barycentre of every 2-face.  The gradient dofs are derivative functionals:
their rows of the Vandermonde matrix come from the order-1 tables of the expansion set
(dual_set.to_riesz, FIAT/dual_set.py:175-205)."""
from . import dual_set, finite_element, functional, polynomial_set


class CubicHermiteDualSet(dual_set.DualSet):
    def __init__(self, ref_el):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        verts = ref_el.get_vertices()
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        nodes = []
        for v in sorted(top[0]):
            jet = [functional.PointEvaluation(ref_el, verts[v])]
            jet += [functional.PointDerivative(ref_el, verts[v], tuple(int(i == d) for i in range(sd))) for d in range(sd)]
            entity_ids[0][v] = list(range(len(nodes), len(nodes) + len(jet)))
            nodes += jet
        if sd > 1:
            for f in sorted(top[2]):
                centre, = ref_el.make_points(2, f, 3)
                entity_ids[2][f] = [len(nodes)]
                nodes.append(functional.PointEvaluation(ref_el, centre))
        super().__init__(nodes, ref_el, entity_ids)


class CubicHermite(finite_element.CiarletElement):
    def __init__(self, ref_el, deg=3):
        assert deg == 3
        poly_set = polynomial_set.ONPolynomialSet(ref_el, 3)
        super().__init__(poly_set, CubicHermiteDualSet(ref_el), 3)
