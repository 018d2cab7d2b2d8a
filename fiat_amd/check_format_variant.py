"""Variant-string parsing for the in-scope families (FIAT/check_format_variant.py
:30-136), including the macro-element splittings ("alfeld", "iso", "iso(k)", "powell-sabin",
"worsey-farin", "powell-sabin(12)").  The spectral point families depend on the absent
third-party recursivenodes package and raise NotImplementedError where points are made."""
import re

from . import macro
from .quadrature import create_quadrature

_CG_POINTS = {"spectral": "gll", "chebyshev": "lgc", "equispaced": "equispaced", "gll": "gll"}
_DG_POINTS = {"spectral": "gl", "chebyshev": "gc", "equispaced": "equispaced",
              "equispaced_interior": "equispaced_interior", "gll": "gll", "gl": "gl"}
_SPLITS = {"iso": macro.IsoSplit, "alfeld": macro.AlfeldSplit, "worsey-farin": macro.WorseyFarinSplit,
           "powell-sabin": macro.PowellSabinSplit, "powell-sabin(12)": macro.PowellSabin12Split}


def parse_lagrange_variant(variant, discontinuous=False, integral=False):
    """-> (splitting, point_variant)."""
    if variant is None:
        variant = "integral" if integral else "equispaced"
    options = variant.replace(" ", "").split(",")
    if len(options) > 2:
        raise ValueError("Illegal variant option")
    if integral:
        table = {"integral": None, "point": "point"}
        point_variant = None
    else:
        table = _DG_POINTS if discontinuous else _CG_POINTS
        point_variant = table["spectral"]
    splitting, iso_degree = None, None
    for raw in options:
        opt = raw.lower()
        if opt in _SPLITS:
            splitting = _SPLITS[opt]
        elif opt.startswith("iso"):
            match = re.match(r"^iso\((\d+)\)$", opt)
            if not match:
                raise ValueError("Illegal variant option")
            iso_degree = int(match.group(1))
        elif opt.startswith("integral"):
            point_variant = opt
        elif opt in table:
            point_variant = table[opt]
        else:
            raise ValueError("Illegal variant option")
    if discontinuous and (splitting is not None or iso_degree is not None) and point_variant in _CG_POINTS.values():
        raise ValueError("Illegal variant. DG macroelements with DOFs on subcell boundaries are not unisolvent.")
    if iso_degree is not None:
        lattice = point_variant or "gll"
        splitting = lambda T: macro.IsoSplit(T, iso_degree, lattice)   # noqa: E731
    return splitting, point_variant


def check_format_variant(variant, degree):
    """-> (splitting, 'point' | 'integral', interpolant_degree)."""
    splitting, variant = parse_lagrange_variant(variant, integral=True)
    if splitting is not None:
        raise NotImplementedError("macro-element splittings are implemented for Lagrange / DiscontinuousLagrange only")
    if variant is None:
        variant = "integral"
    interpolant_degree = None
    match = re.match(r"^integral(?:\((-?\d+)\))?$", variant)
    if match:
        variant = "integral"
        extra, = match.groups()
        interpolant_degree = degree + (int(extra) if extra is not None else 0)
        if interpolant_degree < degree:
            raise ValueError(f"Quadrature degree should be at least {degree}")
    if variant not in {"point", "integral"}:
        raise ValueError('Choose either variant="point" or variant="integral" or variant="integral(q)"')
    return splitting, variant, interpolant_degree


def parse_quadrature_scheme(ref_el, degree, quad_scheme=None):
    """Rule from a ``quad_scheme`` string (FIAT/check_format_variant.py:100-136): comma-separated scheme name
    ("default", "canonical") and/or a splitting ("alfeld", "iso", "powell-sabin", ...) that makes the rule composite."""
    scheme = "default"
    for opt in filter(None, (quad_scheme or "").split(",")):
        if opt in _SPLITS:
            ref_el = _SPLITS[opt](ref_el)
        elif opt.startswith("KMV"):
            raise NotImplementedError("the Kong-Mulder-Veldhuizen lumped rules are out of scope for fiat_amd")
        else:
            scheme = opt
    return create_quadrature(ref_el, degree, scheme)
