"""Raviart-Thomas elements, H(div), on simplices.

Space of degree q = k + 1:  P_k^d + x P~_k.  Degrees of freedom ("integral" variant): on every facet the
component along the facet-area-scaled normal against an orthonormal basis of P_k of the facet, in the cell
all Cartesian components against P_{k-1}.  "point" variant: scaled-normal components at facet lattice
points, Cartesian components at interior lattice points.  Behaviour as FIAT/raviart_thomas.py:17-157
(same nodal basis and numbering); written as dof blocks over fiat_amd/dof_layout.py."""
from . import finite_element, functional
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout, augmented_vector_space


def RTSpace(ref_el, degree):
    return augmented_vector_space(ref_el, degree - 1, lambda p, x: p[:, None, :] * x[None])


def raviart_thomas_dofs(cell, q, variant, moment_degree, scheme):
    lay = DofLayout(cell)
    sd, k = lay.sd, q - 1
    if variant == "integral":
        lay.moments(sd - 1, k if sd > 1 else 0, moment_degree + k, lambda f: [cell.compute_scaled_normal(f)],
                    scheme=scheme)
        if k > 0:
            lay.component_moments(sd, k - 1, moment_degree + k - 1, scheme=scheme)
    else:
        lay.lattice(sd - 1, sd + k, lambda f, pts: [functional.PointScaledNormalEvaluation(cell, f, x) for x in pts])
        if k > 0:
            lay.lattice(sd, sd + k, lambda _, pts: [functional.ComponentPointEvaluation(cell, c, (sd,), x)
                                                    for c in range(sd) for x in pts])
    return lay.dual_set()


class RTDualSet:
    """Constructor-compatible name: ``RTDualSet(ref_el, degree, variant, interpolant_deg, quad_scheme)``."""

    def __new__(cls, ref_el, degree, variant, interpolant_deg, quad_scheme=None):
        return raviart_thomas_dofs(ref_el, degree, variant, interpolant_deg, quad_scheme)


class RaviartThomas(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, moment_degree = check_format_variant(variant, degree)
        super().__init__(RTSpace(ref_el, degree), raviart_thomas_dofs(ref_el, degree, variant, moment_degree, quad_scheme),
                         degree, formdegree=ref_el.get_spatial_dimension() - 1, mapping="contravariant piola")
