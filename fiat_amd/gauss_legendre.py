"""``from FIAT.gauss_legendre import GaussLegendre`` (FIAT/gauss_legendre.py): the class lives in discontinuous_lagrange.py."""
from .discontinuous_lagrange import GaussLegendre  # noqa: F401
