"""Regge elements: symmetric-matrix-valued P_k with continuous tangential-tangential components.

For every sub-entity E of dimension m >= 1 and every edge tangent t of E: t^T u t against an orthonormal
basis of P_{k-m+1}(E) ("integral"), or at the lattice points of E ("point").  Behaviour as FIAT/regge.py:11-79;
written as dof blocks over fiat_amd/dof_layout.py.  Tabulation: ndof * sd * sd rows on the same kernels as
every other coeffs x Dubiner element."""
import numpy

from . import finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout


def regge_dofs(cell, k, variant, moment_degree, scheme):
    lay = DofLayout(cell)
    for m in range(1, lay.sd + 1):
        edges_of = lambda e, m=m: cell.compute_face_edge_tangents(m, e)    # noqa: E731
        if variant == "point":
            lay.lattice(m, k + 2, lambda e, pts, edges_of=edges_of: [functional.PointwiseInnerProductEvaluation(cell, t, t, x)
                                                                      for x in pts for t in edges_of(e)])
        else:
            lay.moments(m, k - m + 1, moment_degree + k - m + 1,
                        lambda e, edges_of=edges_of: [numpy.outer(t, t) for t in edges_of(e)], scheme=scheme,
                        tag="TensorBidirectionalMomentInnerProductEvaluation")
    return lay.dual_set()


class Regge(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=0, variant=None, quad_scheme=None):
        if degree < 0:
            raise ValueError(f"{type(self).__name__} only defined for degree >= 0")
        _, variant, moment_degree = check_format_variant(variant, degree)
        super().__init__(polynomial_set.ONSymTensorPolynomialSet(ref_el, degree),
                         regge_dofs(ref_el, degree, variant, moment_degree, quad_scheme), degree,
                         formdegree=(1, 1), mapping="double covariant piola")
