"""Build-container tool (pytest plugin; not part of the repo's tests, never runs on the GPU box): runs the REFERENCE's unit test
files, where they lie under /root/reference, against the fiat_amd facade -- ``FIAT`` and its sub-modules are aliased to ``fiat_amd``,
names the facade does not have become placeholders that raise NotImplementedError when USED, and the CPU oracle stands in for the
device (tests/host_backend.py).  A gap finder for host-side behaviour:

    PYTHONPATH=tools python -B -m pytest -p no:cacheprovider -p refrun_plugin -c /dev/null --rootdir=build /root/reference/test/FIAT/unit -q

(``-B`` / ``sys.dont_write_bytecode`` and ``-p no:cacheprovider``: the reference tree is read-only by rule, nothing is written into it.)

Round 3: 1385 of the reference's 2 003 unit tests pass this way; the rest need the device (macro elements, fused tensor kernels:
covered by tests/test_gpu_*.py), gem / sympy, or families outside SURVEY section 8.  It found: no ``get_connectivity``, no
``RadauQuadratureLineRule``, no ``make_bubbles``, ``distance_to_point_l1`` without ``entity=``, DG refused on split intervals,
AttributeError instead of NotImplementedError from ``TensorProductElement.get_nodal_basis``, and the missing module names
``FIAT.gauss_legendre`` / ``gauss_lobatto_legendre`` / ``P0`` / ``quadrature_schemes`` and class names ``ReferenceElement``,
``LineExpansionSet`` ..."""
import importlib
import sys
import types

sys.dont_write_bytecode = True      # nothing may be written under /root/reference: no __pycache__ next to its test files

import os  # noqa: E402
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import fiat_amd  # noqa: E402
import host_backend as hb  # noqa: E402
from fiat_amd import runtime  # noqa: E402
import pytest  # noqa: E402


class _Missing:
    def __init__(self, name):
        self._name = name
        self.__name__ = name.split(".")[-1]

    def __call__(self, *a, **k):
        return _Missing(self._name + "()")     # (import-time parametrisations build elements: fail when USED, not when built)

    def __getattr__(self, item):
        raise NotImplementedError(f"{self._name}.{item} is out of scope for fiat_amd")


class _Shim(types.ModuleType):
    def __init__(self, name, target):
        super().__init__(name)
        self._target = target
        self.__path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        t = self.__dict__["_target"]
        if t is not None and hasattr(t, item):
            return getattr(t, item)
        return _Missing(f"{self.__name__}.{item}")


import importlib.abc, importlib.machinery  # noqa: E402


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.startswith("FIAT.") and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        sub = spec.name.split(".", 1)[1]
        try:
            target = importlib.import_module("fiat_amd." + sub)
        except Exception:
            target = None
        return _Shim(spec.name, target)

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Finder())
sys.modules["FIAT"] = _Shim("FIAT", fiat_amd)
for sub in ("reference_element", "quadrature", "quadrature_schemes", "expansions", "polynomial_set", "functional", "dual_set", "finite_element",
            "tensor_product", "macro", "barycentric_interpolation", "jacobi", "lagrange", "discontinuous_lagrange", "nedelec", "raviart_thomas",
            "check_format_variant", "hdivcurl", "enriched", "restricted", "bubble", "orientation_utils", "pointwise_dual", "hierarchical",
            "discontinuous_pc", "P0", "regge", "hellan_herrmann_johnson"):
    try:
        target = importlib.import_module("fiat_amd." + sub)
    except Exception:
        target = None
    if sub == "quadrature_schemes":
        target = importlib.import_module("fiat_amd.quadrature")
    sys.modules["FIAT." + sub] = _Shim("FIAT." + sub, target)


@pytest.fixture(autouse=True)
def _oracle(monkeypatch):
    for name, obj in (("Context", hb._Ctx), ("SimplexPolySet", hb._SimplexPolySet), ("LineLagrange", hb._LineLagrange), ("MacroPolySet", hb._MacroPolySet),
                      ("riesz_assemble", hb._riesz_assemble), ("vandermonde_solve_batch", hb._vandermonde_solve_batch),
                      ("map_points", hb._map_points), ("tables_squared_norm", hb._tables_squared_norm)):
        monkeypatch.setattr(runtime, name, obj)


# module-level code of some test files builds elements at import time: install the CPU stand-ins globally as well
for _name, _obj in (("Context", hb._Ctx), ("SimplexPolySet", hb._SimplexPolySet), ("LineLagrange", hb._LineLagrange), ("MacroPolySet", hb._MacroPolySet),
                    ("riesz_assemble", hb._riesz_assemble), ("vandermonde_solve_batch", hb._vandermonde_solve_batch),
                    ("map_points", hb._map_points), ("tables_squared_norm", hb._tables_squared_norm)):
    setattr(runtime, _name, _obj)
