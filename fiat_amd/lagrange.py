"""Lagrange (CG) elements on simplices: point values on the lattice of every sub-entity -- vertices,
then edges, faces, interior -- over the C0 ("bubble") expansion set with scale 1 on triangles and
tetrahedra, over the primal 1-D Lagrange basis on intervals.  Behaviour as FIAT/lagrange.py:15-88."""
from . import finite_element, functional, polynomial_set
from .barycentric_interpolation import LagrangePolynomialSet, get_lagrange_points
from .check_format_variant import parse_lagrange_variant
from .dof_layout import DofLayout
from .dual_set import DualSet
from .reference_element import LINE


def lagrange_dofs(cell, degree, lattice_family, by_vertices=False):
    """Entities are visited by dimension and number, or -- ``by_vertices`` -- in the lexicographic
    order of their vertex tuples (the numbering of spectral elements)."""
    lay = DofLayout(cell)
    visit = [(dim, e) for dim in sorted(lay.topology) for e in lay.entities(dim)]
    if by_vertices:
        visit.sort(key=lambda de: lay.topology[de[0]][de[1]])
    for dim, e in visit:
        lay.place(dim, e, (functional.PointEvaluation(cell, x)
                           for x in cell.make_points(dim, e, degree, variant=lattice_family)))
    return lay


class LagrangeDualSet(DualSet):
    def __init__(self, ref_el, degree, point_variant="equispaced", sort_entities=False):
        super().__init__(*lagrange_dofs(ref_el, degree, point_variant, sort_entities).parts())


class Lagrange(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant="equispaced", sort_entities=False):
        splitting, lattice_family = parse_lagrange_variant(variant)
        if splitting is not None:       # macro element: nodes and C0 expansion set live on the split cell
            ref_el = splitting(ref_el)
        # (on a split interval the reference keeps the primal 1-D basis piecewise, FIAT/lagrange.py:80-84, to save one basis
        # change's round-off; here the C0 hierarchy of the split interval serves it, as on split triangles and tetrahedra --
        # the same space and nodal basis, coefficients with respect to the other expansion set)
        on_line = ref_el.get_shape() == LINE and not ref_el.is_macrocell()
        dual = LagrangeDualSet(ref_el, degree, lattice_family, sort_entities)
        if on_line:
            space = LagrangePolynomialSet(ref_el, get_lagrange_points(dual))
        else:
            space = polynomial_set.ONPolynomialSet(ref_el, degree, variant="bubble", scale=1)
        super().__init__(space, dual, degree, formdegree=0)


class GaussLobattoLegendre(Lagrange):
    """Nodes at the (recursive) Gauss-Lobatto-Legendre points, entities ordered by their vertices
    (FIAT/gauss_lobatto_legendre.py)."""

    def __init__(self, ref_el, degree):
        super().__init__(ref_el, degree, variant="gll", sort_entities=True)
