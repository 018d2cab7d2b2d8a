"""Nedelec first-kind H(curl) element on triangles and tetrahedra
(FIAT/nedelec.py:17-217): space P_k^d + (homogeneous P_k) x X built by quadrature
projection and an SVD span; dofs = tangential moments on edges, tangential
moments on faces (3-D) and interior moments ("integral" variant)."""
from itertools import chain

import numpy

from . import dual_set, expansions, finite_element, functional, polynomial_set
from .check_format_variant import check_format_variant, parse_quadrature_scheme
from .quadrature import FacetQuadratureRule, create_quadrature


def _vector_subsets(ref_el, k):
    sd = ref_el.get_spatial_dimension()
    vec_Pkp1 = polynomial_set.ONPolynomialSet(ref_el, k + 1, (sd,))
    dims = [expansions.polynomial_dimension(ref_el, j) for j in (k - 1, k, k + 1)]
    return vec_Pkp1, dims


def NedelecSpace2D(ref_el, degree):
    """(P_{degree-1})^2 + P_{degree-1}^hom rot(x)."""
    sd = ref_el.get_spatial_dimension()
    if sd != 2:
        raise ValueError("NedelecSpace2D requires 2d reference element")
    k = degree - 1
    vec_Pkp1, (dimPkm1, dimPk, dimPkp1) = _vector_subsets(ref_el, k)
    vec_Pk = vec_Pkp1.take(list(chain(*(range(i * dimPkp1, i * dimPkp1 + dimPk) for i in range(sd)))))
    Pkp1 = polynomial_set.ONPolynomialSet(ref_el, k + 1)
    PkH = Pkp1.take(list(range(dimPkm1, dimPk)))
    Q = create_quadrature(ref_el, 2 * (k + 1))
    Qpts, Qwts = Q.get_points(), Q.get_weights()
    PkH_at_Qpts = PkH.tabulate(Qpts)[(0,) * sd]
    Pkp1_at_Qpts = Pkp1.tabulate(Qpts)[(0,) * sd]
    rot_x = numpy.array([[0.0, 1.0], [-1.0, 0.0]]) @ Qpts.T
    vals = PkH_at_Qpts[:, None, :] * rot_x[None, :, :]
    coeffs = numpy.dot(vals * Qwts, Pkp1_at_Qpts.T)
    PkHrotX = polynomial_set.PolynomialSet(ref_el, k + 1, k + 1, vec_Pkp1.get_expansion_set(), coeffs)
    return polynomial_set.polynomial_set_union_normalized(vec_Pk, PkHrotX)


def NedelecSpace3D(ref_el, degree):
    """(P_{degree-1})^3 + (P_{degree-1}^hom)^3 x X."""
    sd = ref_el.get_spatial_dimension()
    if sd != 3:
        raise ValueError("NedelecSpace3D requires 3d reference element")
    k = degree - 1
    vec_Pkp1, (dimPkm1, dimPk, dimPkp1) = _vector_subsets(ref_el, k)
    vec_Pk = vec_Pkp1.take(list(chain(*(range(i * dimPkp1, i * dimPkp1 + dimPk) for i in range(sd)))))
    vec_Pke = vec_Pkp1.take(list(chain(*(range(i * dimPkp1 + dimPkm1, i * dimPkp1 + dimPk)
                                         for i in range(sd)))))
    Pkp1 = polynomial_set.ONPolynomialSet(ref_el, k + 1)
    Q = create_quadrature(ref_el, 2 * (k + 1))
    Qpts, Qwts = Q.get_points(), Q.get_weights()
    Pke_at_Qpts = vec_Pke.tabulate(Qpts)[(0,) * sd]
    Pkp1_at_Qpts = Pkp1.tabulate(Qpts)[(0,) * sd]
    cross = numpy.cross(Pke_at_Qpts, Qpts.T[None, :, :], axis=1)
    coeffs = numpy.dot(cross * Qwts, Pkp1_at_Qpts.T)
    PkCrossX = polynomial_set.PolynomialSet(ref_el, k + 1, k + 1, vec_Pkp1.get_expansion_set(), coeffs)
    return polynomial_set.polynomial_set_union_normalized(vec_Pk, PkCrossX)


class NedelecDual(dual_set.DualSet):
    def __init__(self, ref_el, degree, variant, interpolant_deg, quad_scheme):
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        nodes = []
        entity_ids = {dim: {entity: [] for entity in top[dim]} for dim in top}
        if variant == "point":  # tangential point evaluations on edge (and face) lattices
            for e in sorted(top[1]):
                first = len(nodes)
                nodes.extend(functional.PointEdgeTangentEvaluation(ref_el, e, pt) for pt in ref_el.make_points(1, e, degree + 1))
                entity_ids[1][e] = list(range(first, len(nodes)))
            if sd > 2 and degree > 1:
                for f in sorted(top[2]):
                    first = len(nodes)
                    pts = ref_el.make_points(2, f, degree + 1)
                    nodes.extend(functional.PointFaceTangentEvaluation(ref_el, f, k, pt) for k in range(2) for pt in pts)
                    entity_ids[2][f] = list(range(first, len(nodes)))
        # tangential moments against an orthonormal basis on edges (and faces)
        for dim in range(1, sd) if variant == "integral" else ():
            phi_deg = degree - dim
            if phi_deg < 0:
                continue
            facet = ref_el.construct_subelement(dim)
            Q_ref = parse_quadrature_scheme(facet, interpolant_deg + phi_deg, quad_scheme)
            Pqmd = polynomial_set.ONPolynomialSet(facet, phi_deg, (dim,))
            Phis = numpy.transpose(Pqmd.tabulate(Q_ref.get_points())[(0,) * dim], (0, 2, 1))
            for entity in sorted(top[dim]):
                first = len(nodes)
                Q = FacetQuadratureRule(ref_el, dim, entity, Q_ref, avg=True)
                R = numpy.array(ref_el.compute_tangents(dim, entity))
                phis = numpy.transpose(numpy.dot(Phis, R), (0, 2, 1))
                nodes.extend(functional.FrobeniusIntegralMoment(ref_el, Q, phi) for phi in phis)
                entity_ids[dim][entity] = list(range(first, len(nodes)))
        # interior moments against P_{degree-sd}^sd
        phi_deg = degree - sd
        if phi_deg >= 0:
            if interpolant_deg is None:
                interpolant_deg = degree
            cell = ref_el.construct_subelement(sd)
            Q_ref = parse_quadrature_scheme(cell, interpolant_deg + phi_deg, quad_scheme)
            Phis = polynomial_set.ONPolynomialSet(cell, phi_deg).tabulate(Q_ref.get_points())[(0,) * sd]
            for entity in sorted(top[sd]):
                Q = FacetQuadratureRule(ref_el, sd, entity, Q_ref)
                first = len(nodes)
                nodes.extend(functional.IntegralMoment(ref_el, Q, phi, (d,), (sd,))
                             for d in range(sd) for phi in Phis)
                entity_ids[sd][entity] = list(range(first, len(nodes)))
        super().__init__(nodes, ref_el, entity_ids)


class Nedelec(finite_element.CiarletElement):
    def __init__(self, ref_el, degree, variant=None, quad_scheme=None):
        _, variant, interpolant_deg = check_format_variant(variant, degree)
        sd = ref_el.get_spatial_dimension()
        if sd == 3:
            poly_set = NedelecSpace3D(ref_el, degree)
        elif sd == 2:
            poly_set = NedelecSpace2D(ref_el, degree)
        else:
            raise NotImplementedError("Nedelec needs a triangle or a tetrahedron")
        dual = NedelecDual(ref_el, degree, variant, interpolant_deg, quad_scheme)
        super().__init__(poly_set, dual, degree, formdegree=1, mapping="covariant piola")
