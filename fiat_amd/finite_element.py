"""FiniteElement / CiarletElement facade.

Mirrors FIAT/finite_element.py (FiniteElement :20-121, CiarletElement :124-219).
Construction assembles V = dual.to_riesz(P) . coeffs^T and solves V^T X = B on
the device (fx_vandermonde_solve_batch); tabulate() runs the HIP tabulation
kernel.  A singular Vandermonde matrix raises numpy.linalg.LinAlgError, as in the
reference (:151-156)."""
import numpy

from . import runtime
from .polynomial_set import PolynomialSet, mis


class FiniteElement:
    def __init__(self, ref_el, dual, order, formdegree=None, mapping="affine", ref_complex=None):
        self.order = order
        self.formdegree = formdegree
        self.ref_el = ref_el
        self.dual = dual
        self.ref_complex = ref_complex or ref_el
        self._mapping = mapping

    def get_reference_element(self):
        return self.ref_el

    def get_reference_complex(self):
        return self.ref_complex

    def get_dual_set(self):
        return self.dual

    def get_order(self):
        return self.order

    def dual_basis(self):
        return self.dual.get_nodes()

    def entity_dofs(self):
        return self.dual.get_entity_ids()

    def entity_closure_dofs(self):
        return self.dual.get_entity_closure_ids()

    def entity_permutations(self):
        return self.dual.get_entity_permutations()

    def get_formdegree(self):
        return self.formdegree

    def mapping(self):
        return [self._mapping] * self.space_dimension()

    def num_sub_elements(self):
        return 1

    def space_dimension(self):
        return len(self.get_dual_set())

    def tabulate(self, order, points, entity=None):
        raise NotImplementedError("Must be specified in the element subclass of FiniteElement.")

    @staticmethod
    def is_nodal():
        return False

    def is_macroelement(self):
        return self.ref_el is not self.ref_complex


class CiarletElement(FiniteElement):
    def __init__(self, poly_set, dual, order, formdegree=None, mapping="affine", ref_complex=None):
        ref_el = dual.get_reference_element()
        ref_complex = ref_complex or poly_set.get_reference_element()
        super().__init__(ref_el, dual, order, formdegree, mapping, ref_complex)
        if len(poly_set) != len(dual):
            raise ValueError(f"Dimension of function space is {len(poly_set)}, but got {len(dual)} nodes.")

        old_coeffs = poly_set.get_coeffs()
        dualmat = dual.to_riesz(poly_set)
        shp = dualmat.shape
        A = dualmat.reshape((shp[0], -1))
        B = old_coeffs.reshape((shp[0], -1))
        X, V = runtime.vandermonde_solve_batch(A, B, return_V=True)   # LinAlgError if singular
        self.V = V.cpu().numpy()[0]
        new_coeffs = X.cpu().numpy()[0].reshape((shp[0],) + shp[1:])
        self.poly_set = PolynomialSet(poly_set.get_reference_element(), poly_set.get_degree(),
                                      poly_set.get_embedded_degree(), poly_set.get_expansion_set(), new_coeffs)
        if hasattr(poly_set.get_expansion_set(), "device_line"):
            # 1-D Lagrange primal basis: keep the specialised tabulate
            from .barycentric_interpolation import LagrangePolynomialSet
            ps = LagrangePolynomialSet.__new__(LagrangePolynomialSet)
            PolynomialSet.__init__(ps, poly_set.get_reference_element(), poly_set.get_degree(),
                                   poly_set.get_embedded_degree(), poly_set.get_expansion_set(), new_coeffs)
            self.poly_set = ps
        es = poly_set.get_expansion_set()
        self._expansion_variant = getattr(es, "variant", None)
        self._expansion_scale = es.get_scale(poly_set.get_embedded_degree())

    def degree(self):
        return self.poly_set.get_embedded_degree()

    def get_nodal_basis(self):
        return self.poly_set

    def get_coeffs(self):
        return self.poly_set.get_coeffs()

    def device_polyset(self):
        """Device-resident nodal basis for the batched API (fx_tabulate_batch)."""
        return self.poly_set.device_polyset()

    def entity_map(self, entity):
        """(M, b) of the affine map from the coordinates of the reference sub-entity ``(dim, id)`` into this
        element's cell, x = M xi + b (reference_element.py:570-609); None for the cell itself."""
        sd = self.ref_el.get_spatial_dimension()
        if entity is None or entity[0] == sd:
            if entity is not None and entity[1] != 0:
                raise ValueError("a simplex has a single cell")
            return None
        dim, number = entity
        f = self.ref_el.get_entity_transform(dim, number)
        b = numpy.asarray(f(numpy.zeros((1, dim))), dtype=float).reshape(sd)
        M = numpy.zeros((sd, dim))
        for i in range(dim):
            unit = numpy.zeros((1, dim))
            unit[0, i] = 1.0
            M[:, i] = numpy.asarray(f(unit), dtype=float).reshape(sd) - b
        return M, b

    def tabulate(self, order, points, entity=None):
        """{alpha: (ndof, *value_shape, npts)} of all derivatives up to ``order``; ``entity=(dim, id)``: the points
        are given in the coordinates of that reference sub-entity (FIAT/finite_element.py:181-197)."""
        points = numpy.asarray(points, dtype=float)
        emap = self.entity_map(entity)
        if emap is None:
            return self.poly_set.tabulate(points, order)
        sd = self.ref_el.get_spatial_dimension()
        single = points.ndim == 1 and entity[0] > 0
        npts = (len(points) if points.ndim == 2 else 1) if entity[0] == 0 else points.size // entity[0]
        dev = self.tabulate_batch(order, points.reshape(1, npts, entity[0]), entity=entity)
        out = runtime.fetch(dev)[0]
        keys = [a for k in range(order + 1) for a in mis(sd, k)]
        return {a: (numpy.ascontiguousarray(out[t][..., 0]) if single else numpy.ascontiguousarray(out[t]))
                for t, a in enumerate(keys)}

    def tabulate_batch(self, order, points, verts=None, out=None, stream=None, pushforward=False, entity=None):
        """Batched form of tabulate(): points (nreq, npts, sd) [+ per-request cell
        vertices (nreq, sd+1, sd)] -> device tensor (nreq, ntab, ndof, *value_shape, npts)
        with tables in mis() order.  ``pushforward=True`` (needs ``verts``) applies this element's
        mapping() -- affine pull-back, covariant or contravariant Piola -- so that the tables are
        the basis functions ON the physical cells.  ``entity=(dim, id)``: points (nreq, npts, dim) in the
        coordinates of that reference sub-entity, mapped into the cell on the device (fx_map_points)."""
        emap = self.entity_map(entity)
        if emap is not None:
            if verts is not None:
                raise NotImplementedError("sub-entity points with per-request cells: use tabulate_cells(..., entity=)")
            points = runtime.map_points(*emap, points, stream=stream)
        mapping = self._mapping if pushforward else None
        return self.device_polyset().tabulate_batch(order, points, verts=verts, out=out, stream=stream, mapping=mapping)

    def tabulate_cells(self, order, ref_points, verts, out=None, stream=None, entity=None):
        """The quadrature-rule case: ONE point set on this element's reference cell (or, ``entity=(dim, id)``, on one
        of its reference sub-entities: a facet rule), pushed forward (with mapping()) to the physical cells ``verts``
        (nreq, sd+1, sd) -> device tensor (nreq, ntab, ndof, *value_shape, npts).  Same result as
        ``tabulate_batch(order, F_r(ref_points), verts, pushforward=True)``."""
        emap = self.entity_map(entity)
        if emap is not None:
            ref_points = runtime.map_points(*emap, ref_points, stream=stream)
        return self.device_polyset().tabulate_batch_shared(order, ref_points, verts, mapping=self._mapping, out=out,
                                                           stream=stream)

    def value_shape(self):
        return self.poly_set.get_shape()

    def get_num_members(self, arg):
        return self.get_nodal_basis().get_expansion_set().get_num_members(arg)

    @staticmethod
    def is_nodal():
        return True


def entity_support_dofs(elem, entity_dim):
    """{entity id: dofs whose basis functions do not vanish on that entity} for the sub-entities of dimension
    ``entity_dim`` (FIAT/finite_element.py:222-264; FInAT stands on the same computation,
    finat/finiteelementbase.py:85-119): the default rule of degree max(2 degree, 1) on the reference sub-entity, mapped
    onto every entity of that dimension, ONE batched order-0 tabulation of all of them on the device, the squared norm of
    every basis function against the weights on the device (fx_tables_squared_norm), cut at 1e-8."""
    import torch
    from .quadrature import create_quadrature
    cache = elem.__dict__.setdefault("_entity_support_dofs", {})
    if entity_dim in cache:
        return cache[entity_dim]
    ref_el = elem.get_reference_element()
    entity_cell = ref_el.construct_subelement(entity_dim)
    quad = create_quadrature(entity_cell, max(2 * elem.degree(), 1))
    weights = numpy.asarray(quad.get_weights(), dtype=float)
    qpts = numpy.asarray(quad.get_points(), dtype=float).reshape(len(weights), -1)
    ids = list(elem.entity_dofs()[entity_dim].keys())
    eps = 1.e-8
    if hasattr(elem, "entity_map"):
        # simplex elements: every entity is one request of the same batch, the entity transform runs on the device
        sd = ref_el.get_spatial_dimension()
        blocks = []
        for f in ids:
            emap = elem.entity_map((entity_dim, f))
            blocks.append(runtime._as_device(qpts, runtime.Context.get()) if emap is None else runtime.map_points(*emap, qpts))
        pts = torch.stack(blocks).reshape(len(ids), len(weights), sd)
        tabs = elem.tabulate_batch(0, pts)[:, 0]                        # (nent, ndof, *value_shape, npts)
    else:
        # tensor-product elements: the factors' entity transforms differ per entity (tabulate_batch(..., entity=))
        tabs = torch.cat([elem.tabulate_batch(0, qpts[None], entity=(entity_dim, f))[:, 0] for f in ids])
    ints = runtime.fetch(runtime.tables_squared_norm(tabs, weights))
    result = {f: [dof for dof, i in enumerate(ints[k]) if i > eps] for k, f in enumerate(ids)}
    cache[entity_dim] = result
    return result
