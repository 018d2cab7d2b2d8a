"""Tensor products of elements: quadrilateral / hexahedral elements from interval factors, prisms from a
triangle and an interval, H(div) / H(curl)-style products with ONE vector-valued factor, nested products.

Behaviour as FIAT/tensor_product.py (TensorProductElement :26-360, tabulate :231-336, FlattenedDimensions :363-434):
basis function (i, j) of A x B has index i * dim(B) + j, the table of the derivative alpha = (alpha_A, alpha_B) is
the per-point product of A's table alpha_A and B's table alpha_B, the value component rides on the vector-valued
factor, ``entity=((dA, dB), id)`` picks a product of sub-entities.

Two device routes: products of scalar 1-D Lagrange / DG factors (BASELINE config 5, the hexahedron) run on the fused
tensor kernel, which evaluates the 1-D bases itself (fx_tensor_tabulate_batch, optionally from per-request 1-D grid
coordinates); every other product tabulates its two factors with their own kernels and multiplies the tables
(fx_table_outer_batch)."""
import numpy

from . import runtime
from .polynomial_set_util import mis
from .reference_element import LINE, TensorProductCell


def _first_point(node):
    return tuple(node.get_point_dict().keys())[0]


def _is_line_lagrange(element):
    if isinstance(element, (TensorProductElement, FlattenedDimensions)):
        return False
    es = element.get_nodal_basis().get_expansion_set()
    return element.get_reference_element().get_shape() == LINE and hasattr(es, "device_line") and element.value_shape() == ()


def _line_factors(element):
    """The interval factors of a (nested) product of scalar 1-D Lagrange elements, left to right, or None."""
    if isinstance(element, FlattenedDimensions):
        return _line_factors(element.element)
    if isinstance(element, TensorProductElement):
        left, right = _line_factors(element.A), _line_factors(element.B)
        return None if left is None or right is None else left + right
    return [element] if _is_line_lagrange(element) else None


def _unravel(cell, entity_dim, entity_id):
    """Entity numbers of the factors: entities of dimension (dA, dB, ...) are numbered row-major."""
    counts = tuple(len(c.get_topology()[d]) for c, d in zip(cell.cells, entity_dim))
    return tuple(int(i) for i in numpy.unravel_index(entity_id, counts))


class TensorProductElement:
    def __init__(self, A, B):
        self.A, self.B = A, B
        self.ref_el = TensorProductCell(A.get_reference_element(), B.get_reference_element())
        if len(A.value_shape()) + len(B.value_shape()) > 1:
            raise NotImplementedError("tabulate does not support two vector-valued inputs")
        kinds = [f.mapping()[0] for f in (A, B)]
        if kinds[0] != "affine" and kinds[1] != "affine":
            raise ValueError("check tensor product mappings - at least one must be affine")
        self._mapping = kinds[0] if kinds[0] != "affine" else kinds[1]
        self._lines = _line_factors(self)
        fa, fb = (getattr(f, "get_formdegree", lambda: None)() for f in (A, B))
        self.formdegree = None if fa is None or fb is None else fa + fb

    # -- accessors of the reference's FiniteElement interface -----------------------------------------------
    def get_reference_element(self):
        return self.ref_el

    def get_order(self):
        return min(self.A.get_order(), self.B.get_order())

    def get_formdegree(self):
        return self.formdegree

    def degree(self):
        """The reference's ``polydegree`` (FIAT/tensor_product.py:215-219): the maximum over the factors."""
        return max(self.A.degree(), self.B.degree())

    def space_dimension(self):
        return self.A.space_dimension() * self.B.space_dimension()

    def value_shape(self):
        return tuple(self.A.value_shape()) or tuple(self.B.value_shape())

    def mapping(self):
        return [self._mapping] * self.space_dimension()

    def get_coeffs(self):
        raise NotImplementedError("get_coeffs not implemented")

    # a product element has no polynomial set of its own (FIAT/tensor_product.py:221-229, 349-356)
    def get_nodal_basis(self):
        raise NotImplementedError("get_nodal_basis not implemented")

    def dmats(self):
        raise NotImplementedError("dmats not implemented")

    def get_num_members(self, arg):
        raise NotImplementedError("get_num_members not implemented")

    def is_nodal(self):
        """Nodal iff both factors are (FIAT/tensor_product.py:357-360)."""
        return all([self.A.is_nodal(), self.B.is_nodal()])

    def entity_dofs(self):
        """{(dA, dB): {entity: dofs}}: dofs of entity (eA, eB) are i * dim(B) + j over the factors' entity dofs,
        entities numbered row-major (FIAT/tensor_product.py:52-68)."""
        a, b = self.A.entity_dofs(), self.B.entity_dofs()
        nb = self.B.space_dimension()
        out = {}
        for da in a:
            for db in b:
                pairs = [(ea, eb) for ea in a[da] for eb in b[db]]
                out[(da, db)] = {k: [i * nb + j for i in a[da][ea] for j in b[db][eb]] for k, (ea, eb) in enumerate(pairs)}
        return out

    # -- tabulation -----------------------------------------------------------------------------------------
    def device_factors(self):
        return [f.get_nodal_basis().get_expansion_set().device_line() for f in self._lines]

    def _prism_factors(self):
        """(device polynomial set of the triangle factor, device 1-D Lagrange factor) of a prism element, or None:
        A an element on a triangle with a nodal basis over the Dubiner expansion set, B a scalar 1-D Lagrange element
        or the constant on the interval."""
        if not hasattr(self, "_prism"):
            self._prism = None
            A, B = self.A, self.B
            simple = not isinstance(A, (TensorProductElement, FlattenedDimensions)) and not isinstance(B, (TensorProductElement, FlattenedDimensions))
            if simple and A.get_reference_element().get_spatial_dimension() == 2 and A.get_reference_element().is_simplex() \
                    and B.get_reference_element().get_shape() == LINE and B.value_shape() == () and hasattr(A, "device_polyset"):
                line = None
                if _is_line_lagrange(B):
                    line = B.get_nodal_basis().get_expansion_set().device_line()
                elif B.space_dimension() == 1 and B.degree() == 0:             # P0: the constant 1 = Lagrange on one node
                    verts = B.get_reference_element().get_vertices()
                    line = runtime.LineLagrange([0.5 * (verts[0][0] + verts[1][0])])
                psA = A.device_polyset() if line is not None else None
                if psA is not None and isinstance(psA, runtime.SimplexPolySet):
                    self._prism = (psA, line)
        return self._prism

    def _split(self, entity):
        """(entity of A, entity of B, point columns of A, point columns of B)."""
        if entity is None:
            entity = (self.ref_el.get_dimension(), 0)
        dims, number = entity
        ea, eb = _unravel(self.ref_el, dims, number)
        sub = self.ref_el.construct_subelement(dims)
        ca, cb = (c.get_spatial_dimension() for c in sub.cells)
        return (dims[0], ea), (dims[1], eb), ca, cb

    def tabulate_batch(self, order, points, out=None, stream=None, grid=False, entity=None):
        """points (nreq, npts, sd) -> (nreq, ntab, ndof, [vdim,] npts) on the device.  ``grid=True`` (products of 1-D
        Lagrange factors): per-request 1-D coordinates (nreq, nf, q) of a tensor grid instead of explicit points.
        ``entity=((dA, dB), id)``: points in the coordinates of that reference sub-entity."""
        whole = entity is None or tuple(entity[0]) == tuple(self.ref_el.get_dimension())
        fused = self._lines is not None and max(f.space_dimension() for f in self._lines) <= 16   # (its node-count bound)
        if fused and whole and (order <= 2 or grid):                        # (the fused kernel serves orders <= 2)
            return runtime.tensor_tabulate_batch(self.device_factors(), order, points, out=out, stream=stream, grid=grid)
        if grid:
            raise NotImplementedError("grid input is served for products of 1-D Lagrange factors only")
        ctx = runtime.Context.get()
        points = runtime._as_device(points, ctx)
        ea, eb, ca, cb = self._split(entity)
        if points.dim() != 3 or points.shape[2] != ca + cb:
            raise ValueError(f"points must have shape (nreq, npts, {ca + cb}), got {tuple(points.shape)}")
        if whole and order <= 2:                                            # prisms: the fused kernel, where it has an instance
            prism = self._prism_factors()
            if prism is not None:
                res = runtime.prism_tabulate_batch(prism[0], prism[1], order, points, out=out, stream=stream)
                if res is not None:
                    return res
        tabs = []
        for factor, ent, cols in ((self.A, ea, points[..., :ca]), (self.B, eb, points[..., ca:ca + cb])):
            tabs.append(factor.tabulate_batch(order, cols.contiguous(), stream=stream, entity=ent))
        sda, sdb = (f.get_reference_element().get_spatial_dimension() for f in (self.A, self.B))
        return runtime.table_outer(order, sda, sdb, tabs[0], tabs[1], out=out, ctx=ctx, stream=stream)

    def tabulate(self, order, points, entity=None):
        """{alpha: (ndof, [vdim,] npts)} for all derivative multi-indices up to ``order``."""
        _, _, ca, cb = self._split(entity)
        pts = numpy.asarray(points, dtype=float).reshape(-1, ca + cb)
        dev = self.tabulate_batch(order, pts[None], entity=entity)
        host = runtime.fetch(dev)[0]
        sd = self.ref_el.get_spatial_dimension()
        keys = [a for k in range(order + 1) for a in mis(sd, k)]
        return {a: numpy.ascontiguousarray(host[t]) for t, a in enumerate(keys)}

    def dual_basis(self):
        """Point evaluations at the concatenated nodes for products of point-evaluation factors
        (FIAT/tensor_product.py:70-190, scalar x scalar case)."""
        from . import functional
        nodes = []
        for na in self.A.dual_basis():
            for nb in self.B.dual_basis():
                if not (isinstance(na, functional.PointEvaluation) and isinstance(nb, functional.PointEvaluation)):
                    raise NotImplementedError("dual basis of products of non-point-evaluation factors")
                nodes.append(functional.PointEvaluation(self.ref_el, _first_point(na) + _first_point(nb)))
        return nodes


def _dimension_total(dims):
    return sum(_dimension_total(d) if isinstance(d, tuple) else d for d in dims) if isinstance(dims, tuple) else dims


def flat_entity_map(cell):
    """{(flat dimension, flat number): (product dimension tuple, product number)}: the entities of a product cell whose
    dimensions sum to d, taken in the sorted order of their dimension tuples and then by number, are the entities 0, 1, ...
    of dimension d of the quadrilateral / hexahedron (FIAT/reference_element.py:1852-1866)."""
    counters, out = {}, {}
    topology = cell.get_topology()
    for dims in sorted(topology):
        flat = _dimension_total(dims)
        for number in sorted(topology[dims]):
            out[(flat, counters.get(flat, 0))] = (dims, number)
            counters[flat] = counters.get(flat, 0) + 1
    return out


class FlattenedDimensions:
    """A tensor-product element viewed on the flattened quadrilateral / hexahedron: same tables, entities numbered by their
    total dimension (FIAT/tensor_product.py:363-434)."""

    def __init__(self, element):
        from .reference_element import UFCHexahedron, UFCQuadrilateral
        self.element = element
        product = element.get_reference_element()
        dim = product.get_spatial_dimension()
        if dim not in (2, 3):
            raise ValueError("Illegal element dimension %s" % dim)
        # the element lives on the UFC quadrilateral / hexahedron (FIAT/tensor_product.py:373-381); tabulation goes through
        # the product element with the entity numbers mapped back
        self.ref_el = UFCQuadrilateral() if dim == 2 else UFCHexahedron()
        self.unflattening_map = flat_entity_map(product)

    def get_reference_element(self):
        return self.ref_el

    def space_dimension(self):
        return self.element.space_dimension()

    def value_shape(self):
        return self.element.value_shape()

    def degree(self):
        return self.element.degree()

    def mapping(self):
        return self.element.mapping()

    # the rest of the FiniteElement interface the reference's class inherits or forwards (FIAT/tensor_product.py:370-434):
    # the dual nodes are the product element's, order / form degree / nodality too
    def dual_basis(self):
        return self.element.dual_basis()

    def get_order(self):
        return self.element.get_order()

    def get_formdegree(self):
        return self.element.get_formdegree()

    def is_nodal(self):
        return self.element.is_nodal()

    def get_coeffs(self):
        return self.element.get_coeffs()

    def get_nodal_basis(self):
        return self.element.get_nodal_basis()

    def dmats(self):
        return self.element.dmats()

    def get_num_members(self, arg):
        return self.element.get_num_members(arg)

    def entity_dofs(self):
        """{flat dimension: {flat number: dofs}} of the product element's entity dofs."""
        product = self.element.entity_dofs()
        out = {}
        for (flat, number), (dims, entity) in self.unflattening_map.items():
            out.setdefault(flat, {})[number] = product[dims][entity]
        return out

    def _product_entity(self, entity):
        if entity is None:
            entity = (self.ref_el.get_spatial_dimension(), 0)
        return self.unflattening_map[tuple(entity)]

    def tabulate(self, order, points, entity=None):
        return self.element.tabulate(order, points, self._product_entity(entity))

    def tabulate_batch(self, order, points, entity=None, **kwargs):
        return self.element.tabulate_batch(order, points, entity=self._product_entity(entity), **kwargs)
