"""Gopalakrishnan-Lederer-Schoeberl elements of the second kind: trace-free matrix-valued P_k with continuous
normal-tangential components (the stress space of the MCS formulation of Stokes flow).

Facets F: t^T u n_F for every tangent t of F against an orthonormal basis of P_k(F).  Cell: for every facet F in turn,
t^T u n_F for the tangents of F against P_{k-1} of the cell.  Behaviour as
FIAT/gopalakrishnan_lederer_schoberl.py:9-71 (GLSDual, GopalakrishnanLedererSchoberlSecondKind); written as dof blocks
over fiat_amd/dof_layout.py.  The space is polynomial_set.TracelessTensorPolynomialSet; tabulation runs ndof * sd * sd
rows on the same kernels as every other coeffs x Dubiner element."""
import numpy

from . import finite_element, polynomial_set
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout

_TAG = "TensorBidirectionalMomentInnerProductEvaluation"


def gls_dofs(cell, k, scheme):
    lay = DofLayout(cell)
    sd = lay.sd
    facets = lay.entities(sd - 1)

    def frames(f):    # t n_f^T for the tangents of facet f
        n = numpy.asarray(cell.compute_scaled_normal(f), dtype=float)
        return [numpy.outer(t, n) for t in cell.compute_tangents(sd - 1, f)]

    def tests(dim, q):   # orthonormal P_q of the reference entity with unit scale (scale=1 in the reference)
        on = polynomial_set.ONPolynomialSet(cell.construct_subelement(dim), q, scale=1)
        return lambda rule: on.tabulate(rule.get_points())[(0,) * dim]

    lay.moments(sd - 1, k, 2 * k, frames, scheme=scheme, tag=_TAG, tests=tests(sd - 1, k))
    if k >= 1:
        for f in facets:       # facet after facet: all tests x tangents of one facet, then the next facet
            lay.moments(sd, k - 1, 2 * k - 1, lambda _, f=f: frames(f), scheme=scheme, tag=_TAG, tests=tests(sd, k - 1))
    return lay.dual_set()


class GopalakrishnanLedererSchoberlSecondKind(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        splitting, variant, _ = check_format_variant(variant, degree)
        if variant != "integral":
            raise ValueError("GLS elements are defined by integral moments")
        if splitting is not None:
            raise NotImplementedError("macro variants of the GLS elements are out of scope for fiat_amd")
        sd = ref_el.get_spatial_dimension()
        super().__init__(polynomial_set.TracelessTensorPolynomialSet(ref_el, degree), gls_dofs(ref_el, degree, quad_scheme),
                         degree, formdegree=(1, sd - 1), mapping="covariant contravariant piola")
