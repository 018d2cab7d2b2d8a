"""Discontinuous element with nodes at the (recursive) Gauss-Legendre points (FIAT/gauss_legendre.py)."""
from . import discontinuous_lagrange


class GaussLegendre(discontinuous_lagrange.DiscontinuousLagrange):
    def __init__(self, ref_el, degree):
        super().__init__(ref_el, degree, variant="gl")
