"""Discontinuous Lagrange elements on simplices: point values at lattice nodes, every node owned by its cell;
orthonormal Dubiner prime basis (1-D: the primal Lagrange basis).  Two node layouts: families that put nodes
on the cell boundary re-use the CG numbering (vertices, edges, faces, interior); interior families lay one full
lattice per cell.  Degree 0 on a single cell is P0 (barycentre value).  Behaviour as
FIAT/discontinuous_lagrange.py:147-241 and FIAT/P0.py:17-52."""
import numpy

from . import finite_element, functional, polynomial_set
from .barycentric_interpolation import LagrangePolynomialSet, get_lagrange_points
from .check_format_variant import parse_lagrange_variant
from .dof_layout import DofLayout
from .dual_set import DualSet
from .reference_element import LINE, make_lattice

_BOUNDARY_FAMILIES = ("equispaced", "gll", "lgc")


def _cell_owned_nodes(cell, point_lists):
    """Point evaluations at the points of ``point_lists`` = [(owning cell, points)], in that order."""
    lay = DofLayout(cell)
    for owner, points in point_lists:
        lay.place(lay.sd, owner, (functional.PointEvaluation(cell, x) for x in points))
    return lay.parts()


class BrokenLagrangeDualSet(DualSet):
    """The CG nodes in CG order, all handed to cell 0."""

    def __init__(self, ref_el, degree, point_variant="equispaced"):
        top = ref_el.get_topology()
        nodes = [(0, ref_el.make_points(dim, e, degree, variant=point_variant)) for dim in sorted(top) for e in sorted(top[dim])]
        super().__init__(*_cell_owned_nodes(ref_el, nodes))


class DiscontinuousLagrangeDualSet(DualSet):
    """One complete lattice per cell of the (possibly split) reference cell."""

    def __init__(self, ref_el, degree, point_variant="equispaced"):
        cells = ref_el.get_topology()[ref_el.get_dimension()]
        nodes = [(c, make_lattice(ref_el.get_vertices_of_subcomplex(cells[c]), degree, variant=point_variant))
                 for c in sorted(cells)]
        super().__init__(*_cell_owned_nodes(ref_el, nodes))


class P0Dual(DualSet):
    def __init__(self, ref_el):
        centre = tuple(numpy.mean(numpy.asarray(ref_el.get_vertices()), axis=0))
        super().__init__(*_cell_owned_nodes(ref_el, [(0, [centre])]))


class P0(finite_element.CiarletElement):
    def __init__(self, ref_el):
        super().__init__(polynomial_set.ONPolynomialSet(ref_el, 0), P0Dual(ref_el), 0,
                         formdegree=ref_el.get_spatial_dimension())


class DiscontinuousLagrange(finite_element.CiarletElement):
    def __new__(cls, ref_el, degree, variant="equispaced"):
        if degree == 0 and not ref_el.is_macrocell() and parse_lagrange_variant(variant, discontinuous=True)[0] is None:
            return P0(ref_el)
        return super().__new__(cls)

    def __init__(self, ref_el, degree, variant="equispaced"):
        splitting, lattice_family = parse_lagrange_variant(variant, discontinuous=True)
        if splitting is not None:
            ref_el = splitting(ref_el)
        # (split intervals: the orthonormal set of the split interval instead of the piecewise primal 1-D basis, as in lagrange.py)
        on_line = ref_el.get_shape() == LINE and not ref_el.is_macrocell()
        layout = BrokenLagrangeDualSet if lattice_family in _BOUNDARY_FAMILIES else DiscontinuousLagrangeDualSet
        dual = layout(ref_el, degree, lattice_family)
        if on_line:
            space = LagrangePolynomialSet(ref_el, get_lagrange_points(dual))
        else:
            space = polynomial_set.ONPolynomialSet(ref_el, degree)
        super().__init__(space, dual, degree, formdegree=ref_el.get_spatial_dimension())


class GaussLegendre(DiscontinuousLagrange):
    """Discontinuous element with nodes at the (recursive) Gauss-Legendre points (FIAT/gauss_legendre.py)."""

    def __init__(self, ref_el, degree):
        super().__init__(ref_el, degree, variant="gl")
