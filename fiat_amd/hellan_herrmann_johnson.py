"""Hellan-Herrmann-Johnson elements: symmetric-matrix-valued P_k with continuous normal-normal components.

Facets: n^T u n against an orthonormal basis of P_k of the facet.  Cell: n_f^T u n_f for every facet normal
against P_{k-1}, and the mixed products n_a^T u n_b of consecutive facet normals against P_k.  "point"
variant: the same bilinear forms at lattice points (on triangles the interior dofs are the Cartesian upper-triangle components).  Behaviour as
FIAT/hellan_herrmann_johnson.py:11-120; written as dof blocks over fiat_amd/dof_layout.py."""
import numpy

from . import finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant
from .dof_layout import DofLayout

_TAG = "TensorBidirectionalMomentInnerProductEvaluation"


def hhj_dofs(cell, k, variant, moment_degree, scheme):
    lay = DofLayout(cell)
    sd = lay.sd
    facets = lay.entities(sd - 1)
    n = {f: numpy.asarray(cell.compute_scaled_normal(f), dtype=float) for f in facets}
    mixed = [(facets[i + 1], facets[i + 2]) for i in range((sd - 1) * (sd - 2))]   # off-diagonal pairs (3-D only)
    if variant == "point":
        at = functional.PointwiseInnerProductEvaluation
        lay.lattice(sd - 1, k + sd, lambda f, pts: [at(cell, n[f], n[f], x) for x in pts])
        if sd == 2:   # triangles keep Cartesian interior dofs: the upper-triangle components at the interior lattice (:38-46)
            comp = functional.ComponentPointEvaluation
            lay.lattice(sd, k + sd, lambda _, pts: [comp(cell, (i, j), (sd, sd), x) for i in range(sd) for j in range(i, sd) for x in pts])
        else:
            lay.lattice(sd, k + sd, lambda _, pts: [at(cell, n[f], n[f], x) for x in pts for f in facets])
            lay.lattice(sd, k + sd + 1, lambda _, pts: [at(cell, n[a], n[b], x) for x in pts for a, b in mixed])
    else:
        q = moment_degree + k
        lay.moments(sd - 1, k, q, lambda f: [numpy.outer(n[f], n[f])], scheme=scheme, tag=_TAG)
        # the lower-degree tests are the leading members of the degree-k basis (NOT a separate degree k-1 set:
        # a degree-0 expansion set carries a different constant)
        interior = polynomial_set.ONPolynomialSet(cell.construct_subelement(sd), k)
        lower = interior.get_expansion_set().get_num_members(k - 1) if k >= 1 else 0
        lay.moments(sd, k, q, lambda _: [numpy.outer(n[f], n[f]) for f in facets], scheme=scheme, tag=_TAG,
                    tests=lambda rule: interior.tabulate(rule.get_points())[(0,) * sd][:lower])
        lay.moments(sd, k, q, lambda _: [numpy.outer(n[a], n[b]) for a, b in mixed], scheme=scheme, tag=_TAG)
    return lay.dual_set()


class HellanHerrmannJohnson(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=0, variant=None, quad_scheme=None):
        if degree < 0:
            raise ValueError(f"{type(self).__name__} only defined for degree >= 0")
        _, variant, moment_degree = check_format_variant(variant, degree)
        sd = ref_el.get_spatial_dimension()
        super().__init__(polynomial_set.ONSymTensorPolynomialSet(ref_el, degree),
                         hhj_dofs(ref_el, degree, variant, moment_degree, quad_scheme), degree,
                         formdegree=(sd - 1, sd - 1), mapping="double contravariant piola")
