"""Lagrange (CG) element on simplices (FIAT/lagrange.py:15-88): point-evaluation
nodes on the equispaced lattice, entity by entity; prime basis = "bubble"
expansion set with scale 1 (identity coefficients)."""
from . import dual_set, finite_element, functional, polynomial_set
from .barycentric_interpolation import LagrangePolynomialSet, get_lagrange_points
from .check_format_variant import parse_lagrange_variant
from .reference_element import LINE


class LagrangeDualSet(dual_set.DualSet):
    def __init__(self, ref_el, degree, point_variant="equispaced", sort_entities=False):
        top = ref_el.get_topology()
        entities = [(dim, entity) for dim in sorted(top) for entity in sorted(top[dim])]
        if sort_entities:
            entities = [e for _, e in sorted((top[d][i], (d, i)) for d, i in entities)]
        nodes = []
        entity_ids = {dim: {} for dim in top}
        for dim, entity in entities:
            first = len(nodes)
            pts = ref_el.make_points(dim, entity, degree, variant=point_variant)
            nodes.extend(functional.PointEvaluation(ref_el, x) for x in pts)
            entity_ids[dim][entity] = list(range(first, len(nodes)))
        super().__init__(nodes, ref_el, entity_ids)


class Lagrange(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant="equispaced", sort_entities=False):
        splitting, point_variant = parse_lagrange_variant(variant)
        if splitting is not None:       # macro element: the nodes and the C0 expansion set live on the split cell
            ref_el = splitting(ref_el)
        if ref_el.is_macrocell() and ref_el.get_shape() == LINE:
            raise NotImplementedError("macro Lagrange elements on intervals")
        dual = LagrangeDualSet(ref_el, degree, point_variant=point_variant, sort_entities=sort_entities)
        if ref_el.get_shape() == LINE:
            poly_set = LagrangePolynomialSet(ref_el, get_lagrange_points(dual))
        else:
            poly_set = polynomial_set.ONPolynomialSet(ref_el, degree, variant="bubble", scale=1)
        super().__init__(poly_set, dual, degree, formdegree=0)


class GaussLobattoLegendre(Lagrange):
    """Nodes at the (recursive) Gauss-Lobatto-Legendre points, entities sorted by their vertices
    (FIAT/gauss_lobatto_legendre.py)."""

    def __init__(self, ref_el, degree):
        Lagrange.__init__(self, ref_el, degree, variant="gll", sort_entities=True)
