"""Continuous element with nodes at the (recursive) Gauss-Lobatto-Legendre points (FIAT/gauss_lobatto_legendre.py)."""
from . import lagrange


class GaussLobattoLegendre(lagrange.Lagrange):
    def __init__(self, ref_el, degree):
        super().__init__(ref_el, degree, variant="gll", sort_entities=True)
