"""fiat_amd -- MI355X-native batched finite-element tabulator.

Drop-in for the tabulate() hot path of FIAT (firedrakeproject/fiat): the same
class names, constructor signatures and return conventions for the in-scope
families, with every table computed by hand-written HIP kernels for gfx950
through the C ABI in include/fiat_amd.h.  There is no CPU fallback.

    from fiat_amd import Lagrange, ufc_simplex, create_quadrature
    el = Lagrange(ufc_simplex(3), 3)
    tab = el.tabulate(1, pts)                 # {alpha: ndarray}, as FIAT
    dev = el.tabulate_batch(1, pts_batch)     # (nreq, ntab, ndof, npts) on the GPU
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP extension is missing)
from .reference_element import (DefaultLine, DefaultTetrahedron, DefaultTriangle,  # noqa: F401
                                UFCInterval, UFCTetrahedron, UFCTriangle, default_simplex,
                                make_affine_mapping, make_lattice, physical_simplex, ufc_simplex)
from .quadrature import create_quadrature, make_quadrature  # noqa: F401
from .polynomial_set import (ONPolynomialSet, ONSymTensorPolynomialSet, PolynomialSet,  # noqa: F401
                             TracelessTensorPolynomialSet, mis)
from .expansions import ExpansionSet  # noqa: F401
from .macro import (AlfeldSplit, IsoSplit, PowellSabin12Split, PowellSabinSplit,  # noqa: F401
                    WorseyFarinSplit)
from .finite_element import CiarletElement, FiniteElement  # noqa: F401
from .lagrange import GaussLobattoLegendre, Lagrange  # noqa: F401
from .discontinuous_lagrange import DiscontinuousLagrange, GaussLegendre  # noqa: F401
from .P0 import P0  # noqa: F401  (through the sub-module of the same name, as FIAT/__init__.py does: a later ``import fiat_amd.P0``
#                                  finds it loaded and does not rebind ``fiat_amd.P0`` from the class to the module)
from .nedelec import Nedelec  # noqa: F401
from .raviart_thomas import RaviartThomas  # noqa: F401
from .brezzi_douglas_marini import BrezziDouglasMarini  # noqa: F401
from .nedelec_second_kind import NedelecSecondKind  # noqa: F401
from .hermite import CubicHermite  # noqa: F401
from .morley import Morley  # noqa: F401
from .regge import Regge  # noqa: F401
from .hellan_herrmann_johnson import HellanHerrmannJohnson  # noqa: F401
from .gopalakrishnan_lederer_schoberl import GopalakrishnanLedererSchoberlSecondKind  # noqa: F401
from .tensor_product import FlattenedDimensions, TensorProductElement  # noqa: F401
from .batch import Request, tabulate_requests  # noqa: F401

# the element registry of the reference (FIAT/__init__.py:72-131), in-scope subset
supported_elements = {
    "Lagrange": Lagrange,
    "Discontinuous Lagrange": DiscontinuousLagrange,
    "Nedelec 1st kind H(curl)": Nedelec,
    "Raviart-Thomas": RaviartThomas,
    "Brezzi-Douglas-Marini": BrezziDouglasMarini,
    "Nedelec 2nd kind H(curl)": NedelecSecondKind,
    "Hermite": CubicHermite,
    "Morley": Morley,
    "Regge": Regge,
    "Gauss-Lobatto-Legendre": GaussLobattoLegendre,
    "Gauss-Legendre": GaussLegendre,
    "Hellan-Herrmann-Johnson": HellanHerrmannJohnson,
    "Gopalakrishnan-Lederer-Schoberl 2nd kind": GopalakrishnanLedererSchoberlSecondKind,
    "TensorProductElement": TensorProductElement,
    "FlattenedDimensions": FlattenedDimensions,
}

# (FIAT/__init__.py:130-131)
extra_elements = {"P0": P0}
