"""Nedelec elements of the first kind, H(curl), on triangles and tetrahedra.

Space of degree q = k + 1:  P_k^d  +  S_k,  S_k = rotated / crossed homogeneous polynomials
(2-D: p (y, -x);  3-D: p e_i x X).  Degrees of freedom ("integral" variant): on every sub-entity of
dimension m = 1 .. d-1 the tangential components against an orthonormal basis of P_{q-m} of the entity,
in the cell all Cartesian components against P_{q-d}.  "point" variant: tangential components at lattice
points of edges and faces.  Behaviour as FIAT/nedelec.py:17-217 (same nodal basis, same numbering);
written as dof blocks over fiat_amd/dof_layout.py."""
import numpy

from . import finite_element, functional
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout, augmented_vector_space


def _rotate(p, x):                      # (np, nq), (2, nq) -> (np, 2, nq)
    return p[:, None, :] * numpy.stack([x[1], -x[0]])[None]


def _cross_axes(p, x):                  # (np, nq), (3, nq) -> (3 np, 3, nq): p (e_i x X)
    axes = numpy.eye(3)
    swirl = numpy.stack([numpy.cross(axes[i][:, None], x, axis=0) for i in range(3)])
    return (swirl[:, None] * p[None, :, None, :]).reshape(-1, 3, x.shape[1])


def NedelecSpace2D(ref_el, degree):
    if ref_el.get_spatial_dimension() != 2:
        raise ValueError("NedelecSpace2D requires 2d reference element")
    return augmented_vector_space(ref_el, degree - 1, _rotate)


def NedelecSpace3D(ref_el, degree):
    if ref_el.get_spatial_dimension() != 3:
        raise ValueError("NedelecSpace3D requires 3d reference element")
    return augmented_vector_space(ref_el, degree - 1, _cross_axes)


def nedelec_dofs(cell, q, variant, moment_degree, scheme):
    lay = DofLayout(cell)
    sd = lay.sd
    if variant == "point":
        lay.lattice(1, q + 1, lambda e, pts: [functional.PointEdgeTangentEvaluation(cell, e, x) for x in pts])
        if sd == 3 and q > 1:
            lay.lattice(2, q + 1, lambda f, pts: [functional.PointFaceTangentEvaluation(cell, f, side, x)
                                                  for side in range(2) for x in pts])
    else:
        for m in range(1, sd):
            lay.moments(m, q - m, moment_degree + q - m, lambda e, m=m: cell.compute_tangents(m, e),
                        scheme=scheme, frame_major=True)
    if q >= sd:
        lay.component_moments(sd, q - sd, (q if moment_degree is None else moment_degree) + q - sd, scheme=scheme)
    return lay.dual_set()


class NedelecDual:
    """Kept as a constructor-compatible name: ``NedelecDual(ref_el, degree, variant, interpolant_deg, quad_scheme)``."""

    def __new__(cls, ref_el, degree, variant, interpolant_deg, quad_scheme=None):
        return nedelec_dofs(ref_el, degree, variant, interpolant_deg, quad_scheme)


class Nedelec(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, moment_degree = check_format_variant(variant, degree)
        spaces = {2: NedelecSpace2D, 3: NedelecSpace3D}
        sd = ref_el.get_spatial_dimension()
        if sd not in spaces:
            raise NotImplementedError("Nedelec needs a triangle or a tetrahedron")
        super().__init__(spaces[sd](ref_el, degree), nedelec_dofs(ref_el, degree, variant, moment_degree, quad_scheme),
                         degree, formdegree=1, mapping="covariant piola")
