"""TensorProductElement facade (FIAT/tensor_product.py:26-360, tabulate :231-336;
FlattenedDimensions :363-434) for products of 1-D Lagrange elements: quadrilateral
and hexahedral Lagrange/DG elements.  The per-point outer products are formed by
the HIP tensor kernel; nested products (A x B) x C flatten to a list of interval
factors whose basis index is row-major over the factors, exactly the reference's
f1g1, f1g2, ... ordering."""
import numpy

from . import runtime
from .polynomial_set_util import mis
from .reference_element import LINE, TensorProductCell


def _line_factors(element):
    """Interval factors of a (nested) tensor-product element, left to right."""
    if isinstance(element, TensorProductElement):
        return _line_factors(element.A) + _line_factors(element.B)
    if isinstance(element, FlattenedDimensions):
        return _line_factors(element.element)
    es = element.get_nodal_basis().get_expansion_set()
    if element.get_reference_element().get_shape() != LINE or not hasattr(es, "device_line"):
        raise NotImplementedError("fiat_amd tensor products take 1-D Lagrange/DG factors")
    return [element]


class TensorProductElement:
    def __init__(self, A, B):
        self.A, self.B = A, B
        cells = []
        for e in (A, B):
            c = e.get_reference_element()
            cells.extend(c.cells if isinstance(c, TensorProductCell) else [c])
        self.ref_el = TensorProductCell(*cells)
        self._factors = _line_factors(self)
        if any(len(f.value_shape()) for f in self._factors):
            raise NotImplementedError("tabulate does not support vector-valued factors in fiat_amd")
        self.order = max(A.get_order(), B.get_order()) if hasattr(A, "get_order") else None

    def get_reference_element(self):
        return self.ref_el

    def get_order(self):
        return max(f.get_order() for f in self._factors)

    def degree(self):
        return sum(f.degree() for f in self._factors)

    def space_dimension(self):
        return int(numpy.prod([f.space_dimension() for f in self._factors]))

    def value_shape(self):
        return ()

    def mapping(self):
        return ["affine"] * self.space_dimension()

    def get_coeffs(self):
        raise NotImplementedError("get_coeffs not implemented")

    def device_factors(self):
        return [f.get_nodal_basis().get_expansion_set().device_line() for f in self._factors]

    def tabulate(self, order, points, entity=None):
        """{alpha: (ndof, npts)} for all derivative multi-indices up to ``order``."""
        if entity is not None and tuple(entity[0]) != tuple(self.ref_el.get_dimension()):
            raise NotImplementedError("sub-entity tabulation of tensor-product elements")
        nf = len(self._factors)
        pts = numpy.asarray(points, dtype=float).reshape(-1, nf)
        out = runtime.tensor_tabulate_batch(self.device_factors(), order, pts[None]).cpu().numpy()[0]
        keys = [a for k in range(order + 1) for a in mis(nf, k)]
        return {a: numpy.ascontiguousarray(out[t]) for t, a in enumerate(keys)}

    def tabulate_batch(self, order, points, out=None, stream=None, grid=False):
        """Batched: points (nreq, npts, nf) -- or, with grid=True, per-request 1-D
        coordinates (nreq, nf, q) of a tensor grid -- -> (nreq, ntab, ndof, npts) on the device."""
        return runtime.tensor_tabulate_batch(self.device_factors(), order, points, out=out, stream=stream, grid=grid)


class FlattenedDimensions:
    """A tensor-product element viewed on the flattened quadrilateral/hexahedron;
    tabulation is unchanged (FIAT/tensor_product.py:396-407)."""

    def __init__(self, element):
        self.element = element
        self.ref_el = element.get_reference_element()

    def get_reference_element(self):
        return self.ref_el

    def space_dimension(self):
        return self.element.space_dimension()

    def value_shape(self):
        return self.element.value_shape()

    def degree(self):
        return self.element.degree()

    def tabulate(self, order, points, entity=None):
        return self.element.tabulate(order, points, None)

    def tabulate_batch(self, *args, **kwargs):
        return self.element.tabulate_batch(*args, **kwargs)
