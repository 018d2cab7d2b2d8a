"""Macro cells: a simplex split into a simplicial complex, the cells the macro elements live on.

Host-side bookkeeping only (vertex placement, topology, parent/child relations); the tabulation on a
complex -- point binning, per-cell recurrence, scatter into the members of the complex -- runs in the
HIP kernel behind ``fx_macro_tabulate_batch`` (see expansions.py).

Mirrors FIAT/macro.py: ``SplitSimplicialComplex`` (:83-199), ``IsoSplit`` (:202-250),
``PowellSabinSplit`` (:253-304), ``AlfeldSplit`` (:307-320), ``WorseyFarinSplit`` (:323-336),
``PowellSabin12Split`` (:339-378), ``make_topology`` (:59-80), ``MacroQuadratureRule`` (:381-432).  The C^k
macro spaces (``CkPolynomialSet``, HCT ...) are out of scope.  Entity and cell numbering follow the reference exactly
(tests/golden/macro.npz pins vertices, topology, connectivity and cell-node maps).
"""
import math
from itertools import combinations

import numpy

from .reference_element import TRIANGLE, Simplex, lattice_iter, make_affine_mapping, make_lattice


def xy_to_bary(verts, pts):
    """Barycentric coordinates of points with respect to a simplex (rows)."""
    A, b = make_affine_mapping(verts, numpy.eye(len(verts)))
    return numpy.asarray(pts, dtype=float) @ A.T + b


def bary_to_xy(verts, bary):
    return numpy.dot(bary, numpy.asarray(verts, dtype=float))


def invert_cell_topology(T):
    """{dim: {vertex tuple: entity}}."""
    return {dim: {verts: entity for entity, verts in T[dim].items()} for dim in T}


def make_topology(sd, num_verts, edges):
    """Topology of the flag complex of an edge list: a (dim+1)-facet for every dim-facet and every
    lower-numbered vertex adjacent to all of its vertices; facets sorted lexicographically."""
    edges = sorted(set(tuple(e) for e in edges))
    topology = {0: {i: (i,) for i in range(num_verts)}, 1: dict(enumerate(edges))}
    neighbours = {v: set() for v in range(num_verts)}
    for a, b in edges:
        neighbours[a].update((a, b))
        neighbours[b].update((a, b))
    for dim in range(1, sd):
        found = []
        for facet in topology[dim].values():
            fv = set(facet)
            found.extend((v, *facet) for v in range(min(facet)) if fv < neighbours[v])
        topology[dim + 1] = dict(enumerate(sorted(found)))
    return topology


class SimplicialComplex(Simplex):
    """Several simplices glued along facets; geometry queries that need a single cell take the first
    cell touching the entity."""

    def is_simplex(self):
        return False

    def volume(self):
        sd = self.get_spatial_dimension()
        return sum(self.volume_of_subcomplex(sd, k) for k in self.topology[sd])

    def volume_of_subcomplex(self, dim, facet_no):
        v = numpy.asarray(self.get_vertices_of_subcomplex(self.topology[dim][facet_no]))
        if dim == 0:
            return 1.0
        E = v[1:] - v[0]
        return math.sqrt(abs(numpy.linalg.det(E @ E.T))) / math.factorial(dim)

    def compute_normal(self, facet_i, cell=None):
        sd = self.get_spatial_dimension()
        top = self.topology
        fverts = top[sd - 1][facet_i]
        if cell is None:
            cell = next(c for c in sorted(top[sd]) if set(fverts) <= set(top[sd][c]))
        cverts = top[sd][cell]
        opposite = next(i for i, v in enumerate(cverts) if v not in fverts)
        A, _ = make_affine_mapping(self.get_vertices_of_subcomplex(cverts), numpy.eye(sd + 1))
        n = -A[opposite]
        return n / numpy.linalg.norm(n)


class SplitSimplicialComplex(SimplicialComplex):
    """A split of ``parent`` (a simplex, or a coarser split of one)."""

    def __init__(self, parent, vertices, topology):
        self._parent_complex = parent
        root = parent
        while root.get_parent():
            root = root.get_parent()
        self._parent_simplex = root
        sd = root.get_spatial_dimension()
        ptop = root.get_topology()
        pinv = invert_cell_topology(ptop)
        bary = xy_to_bary(root.get_vertices(), vertices)

        # every entity of the complex lies in the interior of exactly one parent entity: the one
        # spanned by the parent vertices with a non-zero barycentric coordinate somewhere on it
        self._child_to_parent = {}
        children = {dim: {e: [] for e in ptop[dim]} for dim in ptop}
        for dim in topology:
            self._child_to_parent[dim] = {}
            for entity, vids in topology[dim].items():
                support = tuple(i for i in range(sd + 1) if numpy.any(abs(bary[list(vids), i]) > 1.e-12))
                pdim = len(support) - 1
                pent = pinv[pdim][support]
                self._child_to_parent[dim][entity] = (pdim, pent)
                children[pdim][pent].append((dim, entity))
        vert_arr = numpy.asarray(vertices, dtype=float)
        self._parent_to_children = {}
        for pdim in children:
            self._parent_to_children[pdim] = {}
            for pent, kids in children[pdim].items():
                if len(kids) > 1:   # lexicographic order of the children's midpoints on the parent entity
                    mid = numpy.array([vert_arr[list(topology[d][e])].mean(axis=0) for d, e in kids])
                    ev = root.get_vertices_of_subcomplex(ptop[pdim][pent])
                    if pdim == sd:
                        lam = xy_to_bary(ev, mid)
                    else:   # coordinates of the entity's vertices, read off the parent's coordinates
                        lam = xy_to_bary(root.get_vertices(), mid)[:, list(ptop[pdim][pent])]
                    kids = [kids[j] for j in numpy.lexsort(lam.T)]
                self._parent_to_children[pdim][pent] = tuple(kids)

        # cell -> global ids of its sub-entities, in the numbering of the parent simplex
        inv = invert_cell_topology(topology)
        self._cell_connectivity = {}
        for cell, cverts in topology[sd].items():
            self._cell_connectivity[cell] = {
                dim: [inv[dim][tuple(cverts[v] for v in ptop[dim][e])] for e in sorted(ptop[dim])] for dim in ptop}
        self._interior_facets = {dim: [e for e in sorted(topology[dim]) if self._child_to_parent[dim][e][0] == sd]
                                 for dim in sorted(topology)}
        super().__init__(root.get_shape(), vertices, topology)

    def get_child_to_parent(self):
        return self._child_to_parent

    def get_parent_to_children(self):
        return self._parent_to_children

    def get_cell_connectivity(self):
        return self._cell_connectivity

    def get_interior_facets(self, dimension):
        return self._interior_facets[dimension]

    def construct_subelement(self, dimension):
        return self.get_parent().construct_subelement(dimension)

    def get_facet_element(self):
        return self.construct_subelement(self.get_spatial_dimension() - 1)

    def is_macrocell(self):
        return True

    def get_parent(self):
        return self._parent_simplex

    def get_parent_complex(self):
        return self._parent_complex

    __eq__ = SimplicialComplex.__eq__     # (geometric equality, reference_element.Cell)

    def __hash__(self):
        return hash((type(self).__name__, self.vertices))


class IsoSplit(SplitSimplicialComplex):
    """Regular refinement: ``degree`` subdivisions of every edge, cells from connecting lattice
    neighbours (degree 2 on a tetrahedron: the inner octahedron is cut along one diagonal)."""

    def __init__(self, ref_el, degree=2, variant=None):
        self.degree = degree
        self.variant = variant
        sd = ref_el.get_spatial_dimension()
        new_verts = make_lattice(ref_el.get_vertices(), degree, variant=variant)
        index = {tuple(alpha): i for i, alpha in enumerate(lattice_iter(0, degree + 1, sd))}
        edges = []
        for alpha in lattice_iter(0, degree, sd):
            # the small simplex hanging off lattice vertex alpha: alpha and alpha + e_i
            corner = [index[tuple(a + b for a, b in zip(alpha, beta))] for beta in lattice_iter(0, 2, sd)]
            edges.extend(tuple(sorted(pair)) for pair in combinations(corner, 2))
        if sd == 3:
            if degree != 2:
                raise NotImplementedError("IsoSplit of a tetrahedron: degree 2 only")
            edges.append(tuple(sorted((index[(1, 0, 0)], index[(0, 1, 1)]))))
        super().__init__(ref_el, tuple(new_verts), make_topology(sd, len(new_verts), edges))

    def construct_subcomplex(self, dimension):
        if dimension == self.get_dimension():
            return self
        sub = self.construct_subelement(dimension)
        return sub if dimension == 0 else IsoSplit(sub, self.degree, self.variant)


class PowellSabinSplit(SplitSimplicialComplex):
    """Barycentres of all entities of dimension >= ``dimension`` become vertices; every entity is
    coned from its barycentre over the (already split) entities of its boundary."""

    def __init__(self, ref_el, dimension=1):
        self.split_dimension = dimension
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        new_verts = [tuple(v) for v in ref_el.get_vertices()]
        below = dimension - 1
        pieces = {below: {e: [tuple(top[below][e])] for e in top[below]}}
        for dim in range(dimension, sd + 1):
            pieces[dim] = {}
            for entity in sorted(top[dim]):
                centre = len(new_verts)
                new_verts.extend(tuple(p) for p in ref_el.make_points(dim, entity, dim + 1))
                boundary = [e for d, e in ref_el.sub_entities[dim][entity] if d == dim - 1]
                pieces[dim][entity] = [(*s, centre) for child in boundary for s in pieces[dim - 1][child]]
        simplices = [s for entity in sorted(pieces[sd]) for s in pieces[sd][entity]]
        topology = {0: {i: (i,) for i in range(len(new_verts))}}
        for dim in range(1, sd):
            facets = [f for s in simplices for f in combinations(s, dim + 1)]
            if dim < dimension:      # unsplit entities keep the parent's numbering
                facets = [tuple(top[dim][e]) for e in sorted(top[dim])] + facets
            topology[dim] = dict(enumerate(dict.fromkeys(facets)))
        topology[sd] = dict(enumerate(simplices))
        parent = ref_el if dimension == sd else PowellSabinSplit(ref_el, dimension=dimension + 1)
        super().__init__(parent, tuple(new_verts), topology)

    def construct_subcomplex(self, dimension):
        if dimension == self.get_dimension():
            return self
        sub = self.get_parent_complex().construct_subcomplex(dimension) \
            if hasattr(self.get_parent_complex(), "construct_subcomplex") else self.construct_subelement(dimension)
        if dimension < self.split_dimension:
            return sub
        return PowellSabinSplit(sub, dimension=self.split_dimension)


def _cached_split(cls, ref_el):
    """One split object of a kind per cell OBJECT (FIAT/macro.py:311-316, 327-332: the variant parser and user code that ask
    for the same split of the same cell share it, so that ``element.ref_complex is AlfeldSplit(cell)``)."""
    cache = ref_el.__dict__.setdefault("_split_cache", {})
    if cls not in cache:
        cache[cls] = object.__new__(cls)
    return cache[cls]


class AlfeldSplit(PowellSabinSplit):
    """Barycentric refinement: every vertex joined to the cell barycentre."""

    def __new__(cls, ref_el):
        return _cached_split(cls, ref_el)

    def __init__(self, ref_el):
        if "_parent_complex" in self.__dict__:   # (the cached object: built already)
            return
        super().__init__(ref_el, dimension=ref_el.get_spatial_dimension())


class WorseyFarinSplit(PowellSabinSplit):
    """Cell and facet barycentres (Powell-Sabin on a triangle, Alfeld on an interval)."""

    def __new__(cls, ref_el):
        return _cached_split(cls, ref_el)

    def __init__(self, ref_el):
        if "_parent_complex" in self.__dict__:
            return
        super().__init__(ref_el, dimension=max(1, ref_el.get_spatial_dimension() - 1))


class PowellSabin12Split(SplitSimplicialComplex):
    """Triangle only: Powell-Sabin plus the edges between the edge midpoints."""

    def __init__(self, ref_el):
        if ref_el.get_shape() != TRIANGLE:
            raise ValueError("PowellSabin12Split needs a triangle")
        verts = ref_el.get_vertices()
        q, h, t = 0.25, 0.5, 1.0 / 3.0
        extra = bary_to_xy(verts, [(t, t, t), (h, h, 0), (h, 0, h), (0, h, h), (h, q, q), (q, h, q), (q, q, h)])
        new_verts = [tuple(v) for v in verts] + [tuple(map(float, p)) for p in extra]
        edges = [(0, 4), (0, 7), (0, 5), (1, 4), (1, 8), (1, 6), (2, 5), (2, 9), (2, 6)]   # half edges, medians
        edges += [(3, k) for k in range(4, 10)]                                             # spokes of the centre
        edges += [(4, 7), (4, 8), (5, 7), (5, 9), (6, 8), (6, 9)]                          # midpoint triangle
        super().__init__(PowellSabinSplit(ref_el), tuple(new_verts), make_topology(2, len(new_verts), edges))

    def construct_subcomplex(self, dimension):
        if dimension == 2:
            return self
        if dimension == 1:
            return AlfeldSplit(self.construct_subelement(1))
        if dimension == 0:
            return self.construct_subelement(0)
        raise ValueError("Illegal dimension")


class MacroQuadratureRule:
    """Composite rule: ``Q_ref`` (a rule on the reference simplex of some dimension) mapped onto every
    entity of that dimension of the complex, or onto the children of the given parent facets; points
    shared by neighbouring entities are merged."""

    def __new__(cls, ref_el, Q_ref, parent_facets=None):
        from .quadrature import FacetQuadratureRule, QuadratureRule
        dim = Q_ref.ref_el.get_spatial_dimension()
        if parent_facets is None:
            facets = sorted(ref_el.get_topology()[dim])
        else:
            children = ref_el.get_parent_to_children()
            facets = [e for pf in parent_facets for d, e in children[dim][pf] if d == dim]
        pts, wts = [], []
        for entity in facets:
            Q = FacetQuadratureRule(ref_el, dim, entity, Q_ref)
            pts.extend(Q.pts)
            wts.extend(Q.wts)
        # merge repeated points (rules with points on the boundary of their simplex)
        sd = ref_el.get_spatial_dimension()
        top = ref_el.get_topology()
        atol = 1e-10
        for cell in sorted(top[sd]):
            bary = xy_to_bary(ref_el.get_vertices_of_subcomplex(top[sd][cell]), pts)
            if numpy.isclose(bary, 0, atol=atol).any():
                order = numpy.lexsort(bary.T)
                upts, uwts, prev = [pts[order[0]]], [wts[order[0]]], order[0]
                for cur in order[1:]:
                    if numpy.allclose(bary[cur], bary[prev], atol=atol):
                        uwts[-1] += wts[cur]
                    else:
                        upts.append(pts[cur])
                        uwts.append(wts[cur])
                    prev = cur
                pts, wts = upts, uwts
        return QuadratureRule(ref_el, tuple(pts), tuple(wts))
