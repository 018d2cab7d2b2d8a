"""``from FIAT.P0 import P0`` (FIAT/P0.py): the class lives in discontinuous_lagrange.py."""
from .discontinuous_lagrange import P0  # noqa: F401
