"""Linear functionals (degrees of freedom) as weighted point evaluations.

Only the kinds the in-scope families need (FIAT/functional.py: Functional
:22-153, PointEvaluation :156-170, ComponentPointEvaluation :173-190,
point derivatives :193-271, point normal / tangent evaluations :499-614, IntegralMoment :286-315, IntegralMomentOf(Normal)Derivative
:318-366, FrobeniusIntegralMoment :369-385).  A functional is data:
``pt_dict = {point: [(weight, component), ...]}`` and
``deriv_dict = {point: [(weight, alpha, component), ...]}``; the arithmetic of
applying it to an expansion set happens on the device (dual_set.to_riesz).
"""
from collections import defaultdict

import numpy


def _derivative_multiindices(sd, *directions):
    """{alpha: weight} of the directional derivative d/ds_1 ... d/ds_m: the tensor product of
    the directions, entries summed per multi-index."""
    S = numpy.asarray(directions[0], dtype=float)
    for s in directions[1:]:
        S = numpy.multiply.outer(S, numpy.asarray(s, dtype=float))
    tau = defaultdict(float)
    for index in numpy.ndindex(S.shape):
        alpha = [0] * sd
        for axis in index:
            alpha[axis] += 1
        tau[tuple(alpha)] += float(S[index])
    return dict(tau)


class Functional:
    def __init__(self, ref_el, target_shape, pt_dict, deriv_dict, functional_type):
        self.ref_el = ref_el
        self.target_shape = target_shape
        self.pt_dict = pt_dict
        self.deriv_dict = deriv_dict
        self.functional_type = functional_type
        self.max_deriv_order = max((sum(wac[1]) for wacs in deriv_dict.values() for wac in wacs), default=0)

    def get_point_dict(self):
        return self.pt_dict

    def get_reference_element(self):
        return self.ref_el

    def get_type_tag(self):
        return self.functional_type

    def tostr(self):
        return self.functional_type


class PointEvaluation(Functional):
    """f -> f(x) for scalar f."""

    def __init__(self, ref_el, x):
        super().__init__(ref_el, (), {tuple(x): [(1.0, ())]}, {}, "PointEval")

    def __call__(self, fn):
        return fn(tuple(self.pt_dict.keys())[0])

    def tostr(self):
        return "u(%s)" % (",".join(str(c) for c in list(self.pt_dict)[0]),)


class ComponentPointEvaluation(Functional):
    """f -> f_comp(x) for f of value shape shp."""

    def __init__(self, ref_el, comp, shp, x):
        # (vector- and tensor-valued functions: comp is an index or a tuple of indices, FIAT/functional.py:173-186)
        if not isinstance(comp, tuple):
            comp = (comp,)
        if len(shp) != len(comp):
            raise ValueError("Component and shape are incompatible")
        if any(i < 0 or i >= n for i, n in zip(comp, shp)):
            raise ValueError("Illegal component")
        self.comp = comp if len(comp) > 1 else comp[0]
        super().__init__(ref_el, shp, {tuple(x): [(1.0, comp)]}, {}, "ComponentPointEval")


class IntegralMoment(Functional):
    """f -> sum_q w_q g(x_q) f_comp(x_q) for a function g tabulated at the rule Q."""

    def __init__(self, ref_el, Q, f_at_qpts, comp=(), shp=()):
        self.Q = Q
        self.f_at_qpts = numpy.asarray(f_at_qpts)
        self.comp = comp
        weights = numpy.multiply(self.f_at_qpts, Q.get_weights())
        pt_dict = {tuple(pt): [(wt, comp)] for pt, wt in zip(Q.get_points(), weights)}
        super().__init__(ref_el, shp, pt_dict, {}, "IntegralMoment")


class FrobeniusIntegralMoment(IntegralMoment):
    """f -> sum_q w_q <G(x_q), f(x_q)> for tensor-valued G of f's value shape."""

    def __init__(self, ref_el, Q, f_at_qpts, nm=None):
        f_at_qpts = numpy.asarray(f_at_qpts)
        shp = tuple(f_at_qpts.shape[:-1])
        if len(Q.pts) != f_at_qpts.shape[-1]:
            raise ValueError("Mismatch in number of quadrature points and values")
        self.Q = Q
        self.comp = slice(None, None)
        self.f_at_qpts = f_at_qpts
        weights = numpy.moveaxis(numpy.multiply(f_at_qpts, Q.get_weights()), -1, 0)
        alphas = list(numpy.ndindex(shp))
        pt_dict = {tuple(pt): [(wt[alpha], alpha) for alpha in alphas]
                   for pt, wt in zip(Q.get_points(), weights)}
        Functional.__init__(self, ref_el, shp, pt_dict, {}, nm or "FrobeniusIntegralMoment")


class PointDerivative(Functional):
    """f -> D^alpha f(x) for scalar f."""

    def __init__(self, ref_el, x, alpha):
        self.alpha = tuple(alpha)
        self.order = sum(self.alpha)
        super().__init__(ref_el, (), {}, {tuple(x): [(1.0, self.alpha, ())]}, "PointDeriv")


class PointDirectionalDerivative(Functional):
    """f -> d f_comp / ds (x)."""

    def __init__(self, ref_el, s, pt, comp=(), shp=(), nm=None):
        sd = ref_el.get_spatial_dimension()
        unit = numpy.eye(sd, dtype=int)
        wacs = [(s[i], tuple(int(a) for a in unit[i]), comp) for i in range(sd)]
        super().__init__(ref_el, shp, {}, {tuple(pt): wacs}, nm or "PointDirectionalDeriv")


class PointNormalDerivative(PointDirectionalDerivative):
    def __init__(self, ref_el, facet_no, pt, comp=(), shp=()):
        super().__init__(ref_el, ref_el.compute_normal(facet_no), pt, comp=comp, shp=shp, nm="PointNormalDeriv")


class PointTangentialDerivative(PointDirectionalDerivative):
    def __init__(self, ref_el, edge_no, pt, comp=(), shp=()):
        super().__init__(ref_el, ref_el.compute_edge_tangent(edge_no), pt, comp=comp, shp=shp, nm="PointTangentialDeriv")


class PointSecondDerivative(Functional):
    """f -> d/ds1 d/ds2 f_comp (x)."""

    def __init__(self, ref_el, s1, s2, pt, comp=(), shp=(), nm=None):
        tau = _derivative_multiindices(ref_el.get_spatial_dimension(), s1, s2)
        super().__init__(ref_el, shp, {}, {tuple(pt): [(w, alpha, comp) for alpha, w in tau.items()]},
                         nm or "PointSecondDeriv")


class PointNormalSecondDerivative(PointSecondDerivative):
    def __init__(self, ref_el, facet_no, pt, comp=(), shp=()):
        n = ref_el.compute_normal(facet_no)
        super().__init__(ref_el, n, n, pt, comp=comp, shp=shp, nm="PointNormalSecondDeriv")


class PointTangentialSecondDerivative(PointSecondDerivative):
    def __init__(self, ref_el, edge_no, pt, comp=(), shp=()):
        t = ref_el.compute_edge_tangent(edge_no)
        super().__init__(ref_el, t, t, pt, comp=comp, shp=shp, nm="PointTangentialSecondDeriv")


class IntegralMomentOfDerivative(Functional):
    """f -> sum_q w_q g(x_q) (d/ds_1 ... d/ds_m f_comp)(x_q)."""

    def __init__(self, ref_el, Q, f_at_qpts, *directions, comp=(), shp=(), nm=""):
        self.Q = Q
        self.f_at_qpts = numpy.asarray(f_at_qpts)
        self.comp = comp
        tau = _derivative_multiindices(ref_el.get_spatial_dimension(), *directions)
        weights = numpy.multiply(self.f_at_qpts, Q.get_weights())
        self.weights = {alpha: weights * t for alpha, t in tau.items()}
        dpt_dict = {tuple(pt): [(wt * t, alpha, comp) for alpha, t in tau.items()]
                    for pt, wt in zip(Q.get_points(), weights)}
        super().__init__(ref_el, shp, {}, dpt_dict, nm or "IntegralMomentOfDerivative")


class IntegralMomentOfNormalDerivative(IntegralMomentOfDerivative):
    """Average over a facet of g times the normal derivative."""

    def __init__(self, ref_el, facet_no, Q_face, f_at_qpts):
        from .quadrature import FacetQuadratureRule
        sd = ref_el.get_spatial_dimension()
        Q = FacetQuadratureRule(ref_el, sd - 1, facet_no, Q_face, avg=True)
        super().__init__(ref_el, Q, f_at_qpts, ref_el.compute_normal(facet_no), nm="IntegralMomentOfNormalDerivative")


class _PointVectorEvaluation(Functional):
    """f -> v . f(pt) for a fixed vector v."""

    def __init__(self, ref_el, v, pt, tag):
        v = numpy.asarray(v, dtype=float)
        self.v = v
        super().__init__(ref_el, v.shape, {tuple(pt): [(v[i], (i,)) for i in range(len(v))]}, {}, tag)


class PointNormalEvaluation(_PointVectorEvaluation):
    """Normal component at a point of a codimension-1 facet."""

    def __init__(self, ref_el, facet_no, pt):
        super().__init__(ref_el, ref_el.compute_normal(facet_no), pt, "PointNormalEval")
        self.n = self.v


class PointScaledNormalEvaluation(_PointVectorEvaluation):
    """Normal component at a point of a facet, the normal scaled by the facet's volume."""

    def __init__(self, ref_el, facet_no, pt):
        super().__init__(ref_el, ref_el.compute_scaled_normal(facet_no), pt, "PointScaledNormalEval")

    def tostr(self):
        return "(u.n)(%s)" % ",".join(map(str, list(self.pt_dict)[0]))


class PointEdgeTangentEvaluation(_PointVectorEvaluation):
    """Tangential component (un-normalised edge tangent) at a point of an edge."""

    def __init__(self, ref_el, edge_no, pt):
        super().__init__(ref_el, ref_el.compute_edge_tangent(edge_no), pt, "PointEdgeTangent")
        self.t = self.v

    def tostr(self):
        return "(u.t)(%s)" % ",".join(map(str, list(self.pt_dict)[0]))


class PointFaceTangentEvaluation(_PointVectorEvaluation):
    """Component along tangent `tno` of a face at a point of that face."""

    def __init__(self, ref_el, face_no, tno, pt):
        super().__init__(ref_el, ref_el.compute_face_tangents(face_no)[tno], pt, "PointFaceTangent")
        self.t = self.v
        self.tno = tno

    def tostr(self):
        return "(u.t%d)(%s)" % (self.tno, ",".join(map(str, list(self.pt_dict)[0])))


class PointwiseInnerProductEvaluation(Functional):
    """u -> v^T u(pt) w for symmetric-matrix-valued u: the Frobenius product with w v^T
    (FIAT/functional.py:639-656)."""

    def __init__(self, ref_el, v, w, pt):
        wvT = numpy.outer(w, v)
        pt_dict = {tuple(pt): [(wvT[idx], idx) for idx in numpy.ndindex(wvT.shape)]}
        super().__init__(ref_el, wvT.shape, pt_dict, {}, "PointwiseInnerProductEval")


class TensorBidirectionalIntegralMoment(FrobeniusIntegralMoment):
    """u -> sum_q w_q f(x_q) v^T u(x_q) w (FIAT/functional.py:659-672)."""

    def __init__(self, ref_el, v, w, Q, f_at_qpts):
        F = numpy.multiply(numpy.outer(v, w)[..., None], f_at_qpts)
        super().__init__(ref_el, Q, F, "TensorBidirectionalMomentInnerProductEvaluation")
