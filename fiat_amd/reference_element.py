"""Reference cells for the tabulate path: simplices (UFC and default), their
topology, lattices and entity maps.  Host-side bookkeeping only (no arithmetic
of the hot path lives here).

Mirrors the parts of FIAT/reference_element.py the path uses: cell classes
(:880-1152), ``make_lattice`` (:79-98, equispaced family), ``lattice_iter``
(:64-76), entity transforms (:570-609), ``make_affine_mapping`` (:1621-1654),
``ufc_simplex``/``default_simplex`` (:1680-1715).
"""
import itertools
import math

import numpy

POINT = 0
LINE = 1
TRIANGLE = 2
TETRAHEDRON = 3
QUADRILATERAL = 11
HEXAHEDRON = 111
TENSORPRODUCT = 99


def lattice_iter(start, finish, depth):
    """Integer tuples (i_1..i_depth) with start <= i_k and sum < finish; the last
    entry varies slowest."""
    if depth == 0:
        yield ()
        return
    if depth == 1:
        for i in range(start, finish):
            yield (i,)
        return
    for last in range(start, finish):
        for head in lattice_iter(start, finish - last, depth - 1):
            yield head + (last,)


def multiindex_equal(d, isum, imin=0):
    """d-tuples with entries >= imin summing to isum."""
    if d <= 0:
        return
    imax = isum - (d - 1) * imin
    if imax < imin:
        return
    for i in range(imin, imax):
        for a in multiindex_equal(d - 1, isum - i, imin=imin):
            yield a + (i,)
    yield (imin,) * (d - 1) + (imax,)


def _interior_family(n):
    """1-D nodes of the "equispaced_interior" family: midpoints of n + 1 equal sub-intervals."""
    return (numpy.arange(n + 1) + 0.5) / (n + 1)


def _equispaced_family(n):
    return numpy.array([0.5]) if n == 0 else numpy.linspace(0.0, 1.0, n + 1)


def _gll_family(n):
    """Gauss-Lobatto-Legendre nodes on [0, 1]: the end points and the roots of P_n' (= Jacobi(1,1) of degree n-1)."""
    if n == 0:
        return numpy.array([0.5])
    if n == 1:
        return numpy.array([0.0, 1.0])
    from scipy.special import roots_jacobi
    return numpy.concatenate([[0.0], 0.5 * (roots_jacobi(n - 1, 1, 1)[0] + 1.0), [1.0]])


def _gl_family(n):
    """Gauss-Legendre nodes on [0, 1]."""
    from scipy.special import roots_jacobi
    return 0.5 * (roots_jacobi(n + 1, 0, 0)[0] + 1.0)


def _lgc_family(n):
    """Chebyshev-Gauss-Lobatto nodes on [0, 1]."""
    return numpy.array([0.5]) if n == 0 else 0.5 * (1.0 - numpy.cos(numpy.pi * numpy.arange(n + 1) / n))


def _gc_family(n):
    """Chebyshev-Gauss nodes on [0, 1]."""
    return 0.5 * (1.0 - numpy.cos(numpy.pi * (2.0 * numpy.arange(n + 1) + 1.0) / (2.0 * n + 2.0)))


_NODE_FAMILIES = {"equispaced": _equispaced_family, "equispaced_interior": _interior_family, "gll": _gll_family,
                  "gl": _gl_family, "lgc": _lgc_family, "gc": _gc_family}


def _recursive_barycentric(alpha, family):
    """Barycentric coordinates of lattice index ``alpha`` by the recursive rule the reference takes from
    ``recursivenodes`` (Isaac, "Recursive, parameter-free, explicitly defined interpolation nodes for
    simplices", 2020, eq. 3.1): on a segment the 1-D nodes; otherwise the weighted mean over the facets
    of the rule one dimension lower, facet i (drop alpha_i) weighted with the 1-D node x_n[n - alpha_i]."""
    n = sum(alpha)
    x = family(n)
    if len(alpha) == 2:
        return numpy.array([x[alpha[0]], x[alpha[1]]])
    acc = numpy.zeros(len(alpha))
    total = 0.0
    for i, ai in enumerate(alpha):
        w = x[n - ai]
        sub = _recursive_barycentric(alpha[:i] + alpha[i + 1:], family)
        acc[:i] += w * sub[:i]
        acc[i + 1:] += w * sub[i:]
        total += w
    return acc / total


def make_lattice(verts, n, interior=0, variant=None):
    """Lattice of points on the simplex spanned by ``verts``; ``interior`` layers next to the boundary are dropped
    (FIAT/reference_element.py:79-98).  Families: "equispaced" (alpha / n), "equispaced_interior", and the spectral
    ones -- "gll", "gl", "lgc", "gc" -- whose 1-D nodes are carried to simplices by the recursive rule that the
    reference takes from the third-party ``recursivenodes`` package (restated from the paper, see
    ``_recursive_barycentric``)."""
    variant = variant or "equispaced"
    if variant not in _NODE_FAMILIES:
        raise ValueError(f"unknown point variant {variant!r}")
    X = numpy.asarray(verts, dtype=float)
    D = len(verts)
    family = _NODE_FAMILIES[variant]
    pts = []
    for alpha in multiindex_equal(D, n, interior):
        if variant == "equispaced":
            bary = numpy.asarray(alpha, dtype=float) / n if n > 0 else numpy.full(D, 1.0 / D)
        elif D == 1:
            bary = numpy.ones(1)
        else:
            bary = _recursive_barycentric(tuple(alpha), family)
        pts.append(tuple(numpy.dot(bary, X)))
    return pts


def make_affine_mapping(xs, ys):
    """(A, b) with A xs[i] + b = ys[i] for the vertices of two simplices."""
    xs = numpy.asarray(xs, dtype=float)
    ys = numpy.asarray(ys, dtype=float)
    if len(xs) != len(ys):
        raise ValueError("simplices with different numbers of vertices")
    # homogeneous coordinates: [A | b] [x; 1] = y  for every vertex
    H = numpy.hstack([xs, numpy.ones((len(xs), 1))])
    if H.shape[0] == H.shape[1]:
        sol = numpy.linalg.solve(H, ys)
    else:
        sol = numpy.linalg.lstsq(H, ys, rcond=None)[0]
    return sol[:-1].T.copy(), sol[-1].copy()


class Cell:
    """Vertices + topology {dim: {entity: vertex ids}}."""

    def __init__(self, shape, vertices, topology):
        self.shape = shape
        self.vertices = tuple(tuple(float(c) for c in v) for v in vertices)
        self.topology = topology
        self.sub_entities = {}
        for dim, ents in topology.items():
            self.sub_entities[dim] = {}
            for e, vids in ents.items():
                vs = frozenset(vids)
                self.sub_entities[dim][e] = sorted((d_, e_) for d_, es_ in topology.items()
                                                   for e_, v_ in es_.items() if vs.issuperset(v_))

    def get_shape(self):
        return self.shape

    def get_connectivity(self):
        """{(d0, d1): [entities of dimension d1 incident to entity 0, 1, ... of dimension d0]}: the sub-entities for d1 < d0,
        the entities it is a sub-entity of otherwise (itself for d1 == d0), each list sorted by number
        (FIAT/reference_element.py:166-185, 221-227; known answers for the tetrahedron and hexahedron in
        test/FIAT/unit/test_reference_element.py:42-57)."""
        if getattr(self, "_connectivity", None) is None:
            top = self.get_topology()
            sets = {d: {e: frozenset(v) for e, v in top[d].items()} for d in top}
            conn = {}
            for d0 in sorted(top):
                for d1 in sorted(top):
                    rows = []
                    for e0 in sorted(top[d0]):
                        if d1 < d0:
                            rows.append(tuple(e1 for e1 in sorted(top[d1]) if sets[d1][e1] <= sets[d0][e0]))
                        else:
                            rows.append(tuple(e1 for e1 in sorted(top[d1]) if sets[d0][e0] <= sets[d1][e1]))
                    conn[(d0, d1)] = rows
            self._connectivity = conn
        return self._connectivity

    def get_vertices(self):
        return self.vertices

    def get_spatial_dimension(self):
        return len(self.vertices[0])

    def get_dimension(self):
        return self.get_spatial_dimension()

    def get_topology(self):
        return self.topology

    def get_vertices_of_subcomplex(self, t):
        return tuple(self.vertices[i] for i in t)

    def get_parent(self):
        return None

    def is_macrocell(self):
        return False

    def is_simplex(self):
        """True for a single simplex only; complexes and product cells say False (FIAT/reference_element.py:327,914)."""
        return False

    def get_parent_complex(self):
        return None

    def is_parent(self, other, strict=False):
        """Is this cell ``other`` or one of the complexes / the simplex it was split from (FIAT/reference_element.py:345-354)?"""
        parent = other.get_parent_complex() if strict else other
        while parent is not None:
            if self == parent:
                return True
            parent = parent.get_parent_complex()
        return False

    def __eq__(self, other):
        """Geometric equality: the same vertex positions and, per dimension, the same entities as vertex tuples -- whatever the
        class (Powell-Sabin with the cell's own split dimension IS the Alfeld split; FIAT/reference_element.py:356-367)."""
        if self is other:
            return True
        if not isinstance(other, Cell) or isinstance(other, TensorProductCell) != isinstance(self, TensorProductCell):
            return False
        A, B = self.get_vertices(), other.get_vertices()
        if len(A) != len(B) or (len(A) and len(A[0]) != len(B[0])) or not numpy.allclose(A, B):
            return False
        atop, btop = self.get_topology(), other.get_topology()
        if set(atop) != set(btop):
            return False
        return all(set(atop[dim].values()) == set(btop[dim].values()) for dim in atop)

    def __ne__(self, other):
        return not self.__eq__(other)

    # refinement order: A > B when A was obtained by splitting B (FIAT/reference_element.py:372-382)
    def __gt__(self, other):
        return other.is_parent(self, strict=True)

    def __lt__(self, other):
        return self.is_parent(other, strict=True)

    def __ge__(self, other):
        return other.is_parent(self, strict=False)

    def __le__(self, other):
        return self.is_parent(other, strict=False)

    def __hash__(self):
        return hash((type(self).__name__, self.shape, self.vertices))


class Simplex(Cell):
    def is_simplex(self):
        return True

    def volume(self):
        v = numpy.asarray(self.vertices)
        sd = self.get_spatial_dimension()
        if sd == 0:
            return 1.0
        return abs(numpy.linalg.det(v[1:] - v[0])) / math.factorial(sd)

    def compute_barycentric_coordinates(self, points, entity=None, rescale=False):
        """Barycentric coordinates of points (npts, sd) with respect to the vertices of ``entity`` = (dim, id) -- default: the
        first cell --, (npts, dim + 1); for an entity of lower dimension: those coordinates of a cell containing it that belong
        to its vertices.  ``rescale``: each coordinate times the height of its vertex over the opposite facet, so that it
        measures a distance (FIAT/reference_element.py:616-644)."""
        points = numpy.asarray(points, dtype=float)
        if points.size == 0:
            return points
        sd = self.get_spatial_dimension()
        top = self.get_topology()
        dim, number = (sd, 0) if entity is None else entity
        cell_vids, keep = top[dim][number], slice(None)
        if dim != sd:     # a cell that contains the entity; keep the coordinates of the entity's vertices
            inside = set(cell_vids)
            cell_vids = next(top[sd][c] for c in sorted(top[sd]) if inside <= set(top[sd][c]))
            keep = [i for i, v in enumerate(cell_vids) if v in inside]
        v = numpy.asarray(self.get_vertices_of_subcomplex(cell_vids), dtype=float)
        # lambda = G [x; 1] with G the inverse of the matrix of homogeneous vertex coordinates
        G = numpy.linalg.inv(numpy.vstack([v.T, numpy.ones(sd + 1)]))[keep]
        lam = points.reshape(-1, sd) @ G[:, :sd].T + G[:, sd]
        if rescale:
            lam = lam / numpy.linalg.norm(G[:, :sd], axis=1)
        return lam.reshape(points.shape[:-1] + (lam.shape[-1],))

    def distance_to_point_l1(self, points, entity=None, rescale=False):
        """0 inside the entity (default: the first cell), otherwise minus the sum of the negative barycentric coordinates
        (FIAT/reference_element.py:651-780: the binning criterion of macro elements, here on the host)."""
        lam = self.compute_barycentric_coordinates(points, entity=entity, rescale=rescale)
        return numpy.maximum(-lam, 0.0).sum(axis=-1)

    def contains_point(self, point, epsilon=0.0, entity=None):
        """FIAT/reference_element.py:782-801."""
        return bool(self.distance_to_point_l1(point, entity=entity) <= epsilon)

    def make_points(self, dim, entity_id, order, variant=None, interior=1):
        """Lattice points in the interior of a sub-entity."""
        if dim == 0:
            return (self.vertices[self.topology[0][entity_id][0]],)
        if 0 < dim <= self.get_spatial_dimension():
            ev = self.get_vertices_of_subcomplex(self.topology[dim][entity_id])
            return make_lattice(ev, order, interior=interior, variant=variant)
        raise ValueError("illegal dimension")

    def compute_tangents(self, dim, i):
        """Un-normalised tangents of entity i: vertex differences to its first vertex."""
        vs = numpy.array(self.get_vertices_of_subcomplex(self.topology[dim][i]))
        return vs[1:] - vs[:1]

    def compute_face_edge_tangents(self, dim, entity_id):
        """All edge tangents of a sub-entity of dimension >= 1: vertex differences v_dest - v_source over the
        vertex pairs source < dest (FIAT/reference_element.py:511-524)."""
        vs = numpy.asarray(self.get_vertices_of_subcomplex(self.topology[dim][entity_id]))
        pairs = [(a, b) for a in range(dim) for b in range(a + 1, dim + 1)]
        return numpy.array([vs[b] - vs[a] for a, b in pairs])

    def compute_edge_tangent(self, edge_i):
        return self.compute_tangents(1, edge_i)[0]

    def compute_face_tangents(self, face_i):
        if self.get_spatial_dimension() != 3:
            raise ValueError("face tangents need a tetrahedron")
        return self.compute_tangents(2, face_i)

    def compute_normal(self, facet_i):
        """Unit outward normal of a codimension-1 facet."""
        sd = self.get_spatial_dimension()
        v = numpy.asarray(self.vertices)
        fverts = self.topology[sd - 1][facet_i]
        opposite = next(i for i in range(sd + 1) if i not in fverts)
        if sd == 1:
            n = v[fverts[0]] - v[opposite]
            return n / numpy.linalg.norm(n)
        # gradient of the barycentric coordinate of the opposite vertex points inwards
        A, _ = make_affine_mapping(v, numpy.eye(sd + 1))
        n = -A[opposite]
        return n / numpy.linalg.norm(n)

    def compute_scaled_normal(self, facet_i):
        """Normal of a codimension-1 facet scaled by the facet volume.  In 2-D and
        3-D the orientation follows the facet's vertex ordering (rotated tangent /
        negative cross product of the tangents), which is what makes the H(div)
        degrees of freedom consistent between neighbouring cells -- it is not
        always the outward normal."""
        sd = self.get_spatial_dimension()
        if sd == 2:
            t, = self.compute_tangents(1, facet_i)
            return numpy.array([t[1], -t[0]])
        if sd == 3:
            t0, t1 = self.compute_tangents(2, facet_i)
            return -numpy.cross(t0, t1)
        return self.compute_normal(facet_i)

    def compute_reference_normal(self, facet_dim, facet_i):
        n = Simplex.compute_normal(self, facet_i)
        return n / numpy.linalg.norm(n, numpy.inf)

    def get_entity_transform(self, dim, entity):
        """Map from the reference sub-entity's coordinates into this cell."""
        sd = self.get_spatial_dimension()
        if dim == sd and len(self.topology[sd]) == 1:   # (a single cell: the identity; the cells of a complex: affine maps below)
            if entity != 0:
                raise ValueError("a simplex has a single cell")
            return lambda x: x
        if dim == 0:
            offset = numpy.asarray(self.vertices[self.topology[0][entity][0]])
            return lambda x: numpy.tile(offset, (len(x), 1)) if numpy.ndim(x) > 1 else offset.copy()
        sub = self.construct_subelement(dim)
        ve = numpy.asarray(sub.get_vertices())
        vc = numpy.asarray(self.get_vertices_of_subcomplex(self.topology[dim][entity]))
        C = numpy.linalg.solve(ve[1:] - ve[:1], vc[1:] - vc[:1])
        offset = vc[0] - ve[0] @ C

        def transform(point):
            return numpy.asarray(point, dtype=float) @ C + offset
        return transform

    def construct_subelement(self, dimension):
        raise NotImplementedError


ReferenceElement = Simplex     # (the reference's name for "a simplex given by vertices and topology", FIAT/reference_element.py:930)


class UFCSimplex(Simplex):
    def construct_subelement(self, dimension):
        return ufc_simplex(dimension)


class _UFCTriangleCell(UFCSimplex):
    def compute_normal(self, facet_i):
        """UFC-consistent unit normal: the rotated edge tangent, not always outward
        (FIAT/reference_element.py:1044-1048)."""
        t = self.compute_tangents(1, facet_i)[0]
        n = numpy.array((t[1], -t[0]))
        return n / numpy.linalg.norm(n)


class _UFCTetrahedronCell(UFCSimplex):
    def compute_normal(self, facet_i):
        """UFC-consistent normal of length 2: minus twice the normalised cross product of the
        face tangents (FIAT/reference_element.py:1148-1152; not a unit vector)."""
        t = self.compute_tangents(2, facet_i)
        n = numpy.cross(t[0], t[1])
        return -2.0 * n / numpy.linalg.norm(n)


class DefaultSimplex(Simplex):
    def construct_subelement(self, dimension):
        return default_simplex(dimension)


class Point(Simplex):
    def __init__(self):
        super().__init__(POINT, ((),), {0: {0: (0,)}})

    def get_spatial_dimension(self):
        return 0

    def construct_subelement(self, dimension):
        return self


def _edges(pairs):
    return {i: p for i, p in enumerate(pairs)}


# UFC numbering: entity i of dimension d-1 is opposite vertex i; edges of the
# tetrahedron are ordered by the pair of vertices they do NOT touch.
_UFC_TOPOLOGY = {
    1: {0: {0: (0,), 1: (1,)}, 1: {0: (0, 1)}},
    2: {0: {0: (0,), 1: (1,), 2: (2,)},
        1: _edges([(1, 2), (0, 2), (0, 1)]),
        2: {0: (0, 1, 2)}},
    3: {0: {0: (0,), 1: (1,), 2: (2,), 3: (3,)},
        1: _edges([(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]),
        2: _edges([(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]),
        3: {0: (0, 1, 2, 3)}},
}
_DEFAULT_TOPOLOGY = {
    1: {0: {0: (0,), 1: (1,)}, 1: {0: (0, 1)}},
    2: {0: {0: (0,), 1: (1,), 2: (2,)},
        1: _edges([(1, 2), (2, 0), (0, 1)]),
        2: {0: (0, 1, 2)}},
    3: {0: {0: (0,), 1: (1,), 2: (2,), 3: (3,)},
        1: _edges([(1, 2), (2, 0), (0, 1), (0, 3), (1, 3), (2, 3)]),
        2: _edges([(1, 3, 2), (2, 3, 0), (3, 1, 0), (0, 1, 2)]),
        3: {0: (0, 1, 2, 3)}},
}
_SHAPES = {1: LINE, 2: TRIANGLE, 3: TETRAHEDRON}


def _unit_vertices(sd, lo, hi):
    verts = [[lo] * sd]
    for i in range(sd):
        v = [lo] * sd
        v[i] = hi
        verts.append(v)
    return verts


def ufc_simplex(spatial_dim):
    """UFC reference simplex: vertices 0 and the unit vectors."""
    if spatial_dim == 0:
        return Point()
    if spatial_dim not in (1, 2, 3):
        raise RuntimeError(f"Can't create UFC simplex of dimension {spatial_dim}.")
    cls = {1: UFCSimplex, 2: _UFCTriangleCell, 3: _UFCTetrahedronCell}[spatial_dim]
    return cls(_SHAPES[spatial_dim], _unit_vertices(spatial_dim, 0.0, 1.0), _UFC_TOPOLOGY[spatial_dim])


def default_simplex(spatial_dim):
    """Default simplex with vertices (-1,..,-1), (1,-1,..), ... used by the expansion sets."""
    if spatial_dim == 0:
        return Point()
    if spatial_dim not in (1, 2, 3):
        raise RuntimeError(f"Can't create default simplex of dimension {spatial_dim}.")
    return DefaultSimplex(_SHAPES[spatial_dim], _unit_vertices(spatial_dim, -1.0, 1.0),
                          _DEFAULT_TOPOLOGY[spatial_dim])


class SymmetricSimplex(Simplex):
    def construct_subelement(self, dimension):
        return symmetric_simplex(dimension)


def symmetric_simplex(spatial_dim):
    """The regular simplex of edge length 2 centred at the origin, UFC topology (FIAT/reference_element.py:966-974,
    1718-1727): vertex 0 = (-1, -1/sqrt 3, -1/sqrt 6), vertex 1 = its mirror image in x, vertex 2 above the first edge's
    midpoint, vertex 3 on the third axis -- truncated to the first ``spatial_dim`` coordinates."""
    if spatial_dim == 0:
        return Point()
    if spatial_dim not in (1, 2, 3):
        raise RuntimeError(f"Can't create symmetric simplex of dimension {spatial_dim}.")
    r3, r6 = math.sqrt(3.0), math.sqrt(6.0)
    full = [(-1.0, -1.0 / r3, -1.0 / r6), (1.0, -1.0 / r3, -1.0 / r6), (0.0, 2.0 / r3, -1.0 / r6), (0.0, 0.0, 3.0 / r6)]
    verts = [v[:spatial_dim] for v in full[:spatial_dim + 1]]
    return SymmetricSimplex(_SHAPES[spatial_dim], verts, _UFC_TOPOLOGY[spatial_dim])


def UFCInterval():
    return ufc_simplex(1)


def UFCTriangle():
    return ufc_simplex(2)


def UFCTetrahedron():
    return ufc_simplex(3)


def DefaultLine():
    return default_simplex(1)


def DefaultTriangle():
    return default_simplex(2)


def DefaultTetrahedron():
    return default_simplex(3)


def physical_simplex(vertices):
    """A simplex with the UFC topology and arbitrary (affine) vertex positions:
    elements may be constructed directly on a physical cell
    (test/finat/test_point_evaluation.py:35-70 exercises that use)."""
    vertices = numpy.asarray(vertices, dtype=float)
    sd = vertices.shape[1]
    return UFCSimplex(_SHAPES[sd], vertices, _UFC_TOPOLOGY[sd])


def _dimension_sum(d):
    return sum(_dimension_sum(x) for x in d) if isinstance(d, tuple) else d


class TensorProductCell(Cell):
    """Product of cells -- intervals (quadrilateral, hexahedron), triangle x interval (prism), or products of
    products: ``cells`` are kept as given, so a nested element (A x B) x C lives on a nested cell whose entity
    dimensions are nested tuples, ((1, 1), 1).  Vertices: concatenated coordinates, the last factor fastest.
    Topology: entities of dimension (d_0, d_1, ...) are products of the factors' entities, numbered row-major over the
    factors' (sorted) entity numbers -- the numbering ``TensorProductElement.tabulate(entity=...)`` unravels."""

    def __init__(self, *cells):
        self.cells = tuple(cells)
        self.shape = TENSORPRODUCT
        self.vertices = tuple(sum((tuple(v) for v in combo), ()) for combo in
                              itertools.product(*[c.get_vertices() for c in cells]))
        self._topology = None
        self.sub_entities = None

    @property
    def topology(self):
        if self._topology is None:
            counts = [len(c.get_vertices()) for c in self.cells]
            tops = [c.get_topology() for c in self.cells]
            topo = {}
            for dims in itertools.product(*[sorted(t, key=repr) for t in tops]):
                entities = {}
                for number, picks in enumerate(itertools.product(*[sorted(t[d]) for t, d in zip(tops, dims)])):
                    corner_sets = [t[d][e] for t, d, e in zip(tops, dims, picks)]
                    entities[number] = tuple(int(numpy.ravel_multi_index(corner, counts))
                                             for corner in itertools.product(*corner_sets))
                topo[dims] = entities
            self._topology = topo
        return self._topology

    def get_topology(self):
        return self.topology

    def get_spatial_dimension(self):
        return sum(c.get_spatial_dimension() for c in self.cells)

    def get_dimension(self):
        return tuple(c.get_dimension() for c in self.cells)

    def construct_subelement(self, dimension):
        return TensorProductCell(*[c.construct_subelement(d) for c, d in zip(self.cells, dimension)])

    def volume(self):
        """Product of the factors' volumes (FIAT/reference_element.py:1247-1249)."""
        return float(numpy.prod([c.volume() for c in self.cells]))

    def _factor_entities(self, dims, entity):
        """Entity ``entity`` of dimension tuple ``dims`` = the product of these entities of the factors (row-major numbering)."""
        counts = tuple(len(c.get_topology()[d]) for c, d in zip(self.cells, dims))
        return tuple(int(i) for i in numpy.unravel_index(entity, counts))

    def get_entity_transform(self, dims, entity):
        """Coordinates on the reference sub-entity of dimensions ``dims`` (the factors' sub-entity coordinates, concatenated)
        -> coordinates in the cell: every factor's own transform on its slice (FIAT/reference_element.py:1221-1245)."""
        picks = self._factor_entities(dims, entity)
        parts = [c.get_entity_transform(d, e) for c, d, e in zip(self.cells, dims, picks)]
        widths = [_dimension_sum(d) for d in dims]

        def transform(point):
            point = numpy.asarray(point, dtype=float)
            out, start = [], 0
            for t, w, c in zip(parts, widths, self.cells):
                piece = point[..., start:start + w]
                if w == 0:   # a vertex of this factor: its coordinates, once per point
                    v = numpy.asarray(t(numpy.zeros((1, 0)))).reshape(-1)
                    piece = numpy.broadcast_to(v, point.shape[:-1] + v.shape)
                else:
                    piece = t(piece)
                out.append(piece)
                start += w
            return numpy.concatenate(out, axis=-1)
        return transform

    def compute_reference_normal(self, facet_dim, facet_i):
        """Unit normal (infinity norm) of a facet: the normal of the one factor whose entity is a facet of it, the other
        factors contribute zeros (FIAT/reference_element.py:1251-1264)."""
        picks = self._factor_entities(facet_dim, facet_i)
        out = []
        for c, d, e in zip(self.cells, facet_dim, picks):
            sd = c.get_spatial_dimension()
            out.append(numpy.asarray(c.compute_reference_normal(d, e), dtype=float) if _dimension_sum(d) == sd - 1
                       else numpy.zeros(sd))
        return numpy.concatenate(out)

    def distance_to_point_l1(self, point, rescale=False):
        """Sum of the factors' distances to their slices of the point (FIAT/reference_element.py:1291-1301)."""
        point = numpy.asarray(point, dtype=float)
        total, start = 0.0, 0
        for c in self.cells:
            n = c.get_spatial_dimension()
            total = total + c.distance_to_point_l1(point[..., start:start + n], rescale=rescale)
            start += n
        return total

    def contains_point(self, point, epsilon=0.0):
        """FIAT/reference_element.py:1266-1289."""
        return bool(self.distance_to_point_l1(point) <= epsilon)

    def flat_cells(self):
        """The simplex factors, left to right, with nesting removed."""
        out = []
        for c in self.cells:
            out.extend(c.flat_cells() if isinstance(c, TensorProductCell) else [c])
        return out

    def is_simplex(self):
        return False

    def __eq__(self, other):
        return isinstance(other, TensorProductCell) and self.cells == other.cells

    def __hash__(self):
        return hash(("TensorProductCell", self.cells))


class Hypercube(Cell):
    """A product of intervals seen as ONE cell: the entities of the product whose dimension tuples sum to d, taken in the
    sorted order of those tuples and then by number, are the entities 0, 1, ... of dimension d (FIAT/reference_element.py:
    1420-1534, flatten_entities :1830-1839, compute_unflattening_map :1854-1866)."""

    def __init__(self, dimension, product):
        self.dimension = dimension
        self.product = product
        self.unflattening_map = {}
        topology, counters = {}, {}
        ptop = product.get_topology()
        for dims in sorted(ptop):
            flat = _dimension_sum(dims)
            for number in sorted(ptop[dims]):
                i = counters.get(flat, 0)
                counters[flat] = i + 1
                topology.setdefault(flat, {})[i] = ptop[dims][number]
                self.unflattening_map[(flat, i)] = (dims, number)
        super().__init__({2: QUADRILATERAL, 3: HEXAHEDRON}[dimension], product.get_vertices(), topology)

    def get_dimension(self):
        return self.get_spatial_dimension()

    def construct_subelement(self, dimension):
        sd = self.get_spatial_dimension()
        if dimension > sd:
            raise ValueError(f"Invalid dimension: {(dimension,)}")
        if dimension == sd:
            return self
        return flatten_reference_cube(self.product.construct_subelement((dimension,) + (0,) * (len(self.product.cells) - 1)))

    def get_entity_transform(self, dim, entity_i):
        return self.product.get_entity_transform(*self.unflattening_map[(dim, entity_i)])

    def volume(self):
        return self.product.volume()

    def compute_reference_normal(self, facet_dim, facet_i):
        assert facet_dim == self.get_spatial_dimension() - 1
        return self.product.compute_reference_normal(*self.unflattening_map[(facet_dim, facet_i)])

    def contains_point(self, point, epsilon=0):
        return self.product.contains_point(point, epsilon=epsilon)

    def distance_to_point_l1(self, point, rescale=False):
        return self.product.distance_to_point_l1(point, rescale=rescale)

    def flat_cells(self):
        return self.product.flat_cells()

    def __gt__(self, other):
        return self.product > other

    def __lt__(self, other):
        return self.product < other

    def __ge__(self, other):
        return self.product >= other

    def __le__(self, other):
        return self.product <= other

    def __hash__(self):
        return hash((type(self).__name__, self.shape, self.vertices))


class UFCHypercube(Hypercube):
    """[0, 1]^d, vertices in lexicographic order (FIAT/reference_element.py:1536-1560)."""

    def __init__(self, dim):
        super().__init__(dim, TensorProductCell(*[ufc_simplex(1)] * dim))

    def construct_subelement(self, dimension):
        sd = self.get_spatial_dimension()
        if dimension > sd:
            raise ValueError(f"Invalid dimension: {dimension}")
        return self if dimension == sd else ufc_hypercube(dimension)


class UFCQuadrilateral(UFCHypercube):
    def __init__(self):
        super().__init__(2)


class UFCHexahedron(UFCHypercube):
    def __init__(self):
        super().__init__(3)


def ufc_hypercube(spatial_dim):
    """Point, UFC interval, quadrilateral or hexahedron (FIAT/reference_element.py:1657-1677)."""
    if spatial_dim == 0:
        return Point()
    if spatial_dim == 1:
        return ufc_simplex(1)
    if spatial_dim == 2:
        return UFCQuadrilateral()
    if spatial_dim == 3:
        return UFCHexahedron()
    raise RuntimeError(f"Can't create UFC hypercube of dimension {spatial_dim}.")


def is_ufc(cell):
    """FIAT/reference_element.py:1778-1787."""
    if isinstance(cell, (Point, UFCSimplex, UFCHypercube)) and not isinstance(cell, SymmetricSimplex):
        return type(cell).__name__ != "UFCSimplex" or numpy.allclose(cell.get_vertices(), ufc_simplex(cell.get_spatial_dimension()).get_vertices())
    if isinstance(cell, TensorProductCell):
        return all(is_ufc(c) for c in cell.cells)
    return False


def is_hypercube(cell):
    """FIAT/reference_element.py:1790-1799: hypercubes, intervals, and products of them."""
    if isinstance(cell, Hypercube) or (isinstance(cell, Simplex) and cell.get_shape() == LINE and not cell.is_macrocell()):
        return True
    if isinstance(cell, TensorProductCell):
        return all(is_hypercube(c) for c in cell.cells)
    return False


def flatten_reference_cube(ref_el):
    """A product of intervals (or of hypercubes and intervals) as the hypercube of its dimension; points, intervals and
    hypercubes as they are (FIAT/reference_element.py:1812-1827)."""
    if ref_el.get_spatial_dimension() <= 1:
        return ref_el
    if isinstance(ref_el, TensorProductCell):
        if is_ufc(ref_el):
            return ufc_hypercube(ref_el.get_spatial_dimension())
        return Hypercube(ref_el.get_spatial_dimension(), ref_el)
    if is_hypercube(ref_el):
        return ref_el
    raise TypeError("Can't flatten cell of type %s" % type(ref_el).__name__)


def ufc_cell(cell):
    """Reference cell of a cell name ("interval", "triangle", ..., "quadrilateral", "a * b") or of an object with a
    ``cellname`` (FIAT/reference_element.py:1730-1755)."""
    celltype = cell if isinstance(cell, str) else cell.cellname
    if " * " in celltype:
        return TensorProductCell(*(ufc_cell(c) for c in celltype.split(" * ")))
    names = {"vertex": lambda: ufc_simplex(0), "interval": lambda: ufc_simplex(1), "triangle": lambda: ufc_simplex(2),
             "tetrahedron": lambda: ufc_simplex(3), "quadrilateral": lambda: ufc_hypercube(2), "hexahedron": lambda: ufc_hypercube(3)}
    if celltype not in names:
        raise RuntimeError(f"Don't know how to create UFC cell of type {str(celltype)}")
    return names[celltype]()
