"""``from FIAT.gauss_lobatto_legendre import GaussLobattoLegendre`` (FIAT/gauss_lobatto_legendre.py): the class lives in lagrange.py."""
from .lagrange import GaussLobattoLegendre  # noqa: F401
