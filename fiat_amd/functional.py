"""Linear functionals (degrees of freedom) as weighted point evaluations.

Only the kinds the in-scope families need (FIAT/functional.py: Functional
:22-153, PointEvaluation :156-170, ComponentPointEvaluation :173-190,
IntegralMoment :286-315, FrobeniusIntegralMoment :369-385).  A functional is
data: ``pt_dict = {point: [(weight, component), ...]}``; the arithmetic of
applying it to an expansion set happens on the device (dual_set.to_riesz).
"""
import numpy


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
        if len(shp) != 1:
            raise ValueError("Illegal shape")
        if comp < 0 or comp >= shp[0]:
            raise ValueError("Illegal component")
        self.comp = comp
        super().__init__(ref_el, shp, {tuple(x): [(1.0, (comp,))]}, {}, "ComponentPointEval")


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
