"""Hsieh-Clough-Tocher C1 macro element on the barycentric (Alfeld) split of a triangle (FIAT/hct.py:19-88).

Prime basis: the C1 piecewise polynomials of the given degree on the split (``macro.CkPolynomialSet`` over the C0
macro expansion set, super-smooth C^(degree-1) at the barycentre); nodes on the parent triangle: first-order jets
at the vertices, moments of the normal derivative (and of tangential derivatives of higher degree) on the edges,
interior moments for degree >= 4.  ``reduced=True`` (degree 3): normal derivatives constrained against P2 on
each edge.  Riesz assembly (point derivatives and derivative moments of the macro expansion set, evaluated with
non-unique binning on the vertices), Vandermonde solve and tabulation run on the device (macro kernels)."""
from . import dual_set, finite_element, macro, polynomial_set
from .check_format_variant import parse_quadrature_scheme
from .functional import (IntegralMoment, IntegralMomentOfDerivative, IntegralMomentOfNormalDerivative,
                         PointDerivative, PointEvaluation)
from .jacobi import eval_jacobi_batch, eval_jacobi_deriv_batch
from .quadrature import FacetQuadratureRule
from .reference_element import TRIANGLE, ufc_simplex


class HCTDualSet(dual_set.DualSet):
    def __init__(self, ref_complex, degree, reduced=False, quad_scheme=None):
        if reduced and degree != 3:
            raise ValueError("Reduced HCT only defined for degree = 3")
        if degree < 3:
            raise ValueError("HCT only defined for degree >= 3")
        ref_el = ref_complex.get_parent()
        if ref_el.get_shape() != TRIANGLE:
            raise ValueError("HCT only defined on triangles")
        top = ref_el.get_topology()
        verts = ref_el.get_vertices()
        sd = ref_el.get_spatial_dimension()
        entity_ids = {dim: {entity: [] for entity in sorted(top[dim])} for dim in sorted(top)}
        nodes = []

        def add(dim, entity, new):
            entity_ids[dim][entity].extend(range(len(nodes), len(nodes) + len(new)))
            nodes.extend(new)

        for v in sorted(top[0]):       # value and gradient at the vertices
            add(0, v, [PointEvaluation(ref_el, verts[v])] +
                [PointDerivative(ref_el, verts[v], alpha) for alpha in polynomial_set.mis(sd, 1)])

        k = 2 if reduced else degree - 3
        edge = ufc_simplex(1)
        Q_ref = parse_quadrature_scheme(edge, degree - 1 + k, quad_scheme)
        xref = 2.0 * Q_ref.get_points() - 1.0          # edge coordinate in (-1, 1)
        if reduced:
            f_at_qpts = eval_jacobi_batch(0, 0, k, xref)[k]
            for e in sorted(top[1]):
                add(1, e, [IntegralMomentOfNormalDerivative(ref_el, e, Q_ref, f_at_qpts)])
        else:
            phis = eval_jacobi_batch(1, 1, k, xref)
            dphis = 2 * eval_jacobi_deriv_batch(1, 1, k, xref)
            for e in sorted(top[1]):
                Q = FacetQuadratureRule(ref_el, 1, e, Q_ref, avg=True)
                n = ref_el.compute_normal(e)
                add(1, e, [IntegralMomentOfDerivative(ref_el, Q, phi, n) for phi in phis] +
                    [IntegralMoment(ref_el, Q, dphi) for dphi in dphis[1:]])
            q = degree - 4
            if q >= 0:                  # interior moments against P_q, composite rule on the split
                Q = parse_quadrature_scheme(ref_complex, degree + q, quad_scheme)
                phis = polynomial_set.ONPolynomialSet(ref_el, q, scale=1).tabulate(Q.get_points())[(0,) * sd]
                phis = phis * (1.0 / ref_el.volume())
                add(sd, 0, [IntegralMoment(ref_el, Q, phi) for phi in phis])
        super().__init__(nodes, ref_el, entity_ids)


class HsiehCloughTocher(finite_element.CiarletElement):
    def __init__(self, ref_el, degree=3, reduced=False, quad_scheme=None):
        ref_complex = macro.AlfeldSplit(ref_el)
        dual = HCTDualSet(ref_complex, degree, reduced=reduced, quad_scheme=quad_scheme)
        poly_set = macro.CkPolynomialSet(ref_complex, degree, order=1, vorder=degree - 1, variant="bubble")
        super().__init__(poly_set, dual, degree, formdegree=0)
