"""Nedelec elements of the second kind, H(curl), on triangles and tetrahedra: the full space P_k^d.

On every sub-entity E of dimension m = 1 .. d: moments against the vector polynomials of degree k - m + 1 of
the reference entity -- all of P_k on edges, Raviart-Thomas RT_{k-m+1} on faces and in the cell -- pulled to E
contravariantly.  "point" variant: tangential point values on the edges instead of the edge moments.
Behaviour as FIAT/nedelec_second_kind.py:20-222; written as dof blocks over fiat_amd/dof_layout.py."""
from . import finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout


def _test_fields(entity_cell, m, degree, variant):
    if m == 1:
        return polynomial_set.ONPolynomialSet(entity_cell, degree, (1,))
    from .raviart_thomas import RaviartThomas
    return RaviartThomas(entity_cell, degree, variant).get_nodal_basis()


def n2curl_dofs(cell, k, variant, moment_degree, scheme):
    lay = DofLayout(cell)
    if moment_degree is None:
        moment_degree = k
    first = 1
    if variant == "point":
        lay.lattice(1, k + 2, lambda e, pts: [functional.PointEdgeTangentEvaluation(cell, e, x) for x in pts])
        first = 2
    for m in range(first, lay.sd + 1):
        deg = k - m + 1
        if deg >= 1:
            space = _test_fields(cell.construct_subelement(m), m, deg, variant)
            lay.field_moments(m, lambda x, space=space, m=m: space.tabulate(x)[(0,) * m], moment_degree + deg,
                              "contravariant", scheme=scheme)
    return lay.dual_set()


class NedelecSecondKind(finite_element.CiarletElement):
    """N2curl_k, k >= 1; variant in {None, "integral", "integral(q)", "point"}."""

    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, moment_degree = check_format_variant(variant, degree)
        assert degree >= 1, "Second kind Nedelecs start at 1!"
        d = ref_el.get_spatial_dimension()
        assert d in (2, 3), "Second kind Nedelecs only implemented in 2/3D."
        super().__init__(polynomial_set.ONPolynomialSet(ref_el, degree, (d,)),
                         n2curl_dofs(ref_el, degree, variant, moment_degree, quad_scheme), degree,
                         formdegree=1, mapping="covariant piola")
