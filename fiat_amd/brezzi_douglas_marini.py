"""Brezzi-Douglas-Marini elements, H(div), on triangles and tetrahedra: the full space P_k^d.

Facets: the scaled-normal component against an orthonormal basis of P_k of the facet ("point": at lattice
points).  Cell (k > 1): moments against the first-kind Nedelec basis of degree k - 1, pulled to the cell
covariantly.  Behaviour as FIAT/brezzi_douglas_marini.py:15-112; written as dof blocks over
fiat_amd/dof_layout.py (same coeffs x Dubiner kernels)."""
from . import finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout


def bdm_dofs(cell, k, variant, moment_degree, scheme):
    from .nedelec import Nedelec
    lay = DofLayout(cell)
    sd = lay.sd
    if variant == "integral":
        lay.moments(sd - 1, k, moment_degree + k, lambda f: [cell.compute_scaled_normal(f)], scheme=scheme)
    else:
        lay.lattice(sd - 1, sd + k, lambda f, pts: [functional.PointScaledNormalEvaluation(cell, f, x) for x in pts])
    if k > 1:
        inner = Nedelec(cell.construct_subelement(sd), k - 1, variant)
        lay.field_moments(sd, lambda x: inner.tabulate(0, x)[(0,) * sd],
                          (k if moment_degree is None else moment_degree) + k - 1, "covariant", scheme=scheme)
    return lay.dual_set()


class BrezziDouglasMarini(finite_element.CiarletElement):
    """BDM_k, k >= 1; variant in {None, "integral", "integral(q)", "point"}."""

    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, moment_degree = check_format_variant(variant, degree)
        if degree < 1:
            raise Exception("BDM_k elements only valid for k >= 1")
        sd = ref_el.get_spatial_dimension()
        if sd not in (2, 3):
            raise NotImplementedError("BrezziDouglasMarini needs a triangle or a tetrahedron")
        super().__init__(polynomial_set.ONPolynomialSet(ref_el, degree, (sd,)),
                         bdm_dofs(ref_el, degree, variant, moment_degree, quad_scheme), degree,
                         formdegree=sd - 1, mapping="contravariant piola")
