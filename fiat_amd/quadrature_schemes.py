"""``from FIAT.quadrature_schemes import create_quadrature`` (FIAT/quadrature_schemes.py:46-106): the function lives in quadrature.py."""
from .quadrature import create_quadrature  # noqa: F401
