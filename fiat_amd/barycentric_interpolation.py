"""1-D Lagrange bases as expansion sets (identity coefficients), evaluated on the
device by the second barycentric formula.

Mirrors FIAT/barycentric_interpolation.py: get_lagrange_points (:13-19), make_dmat
(:50-59), LagrangeLineExpansionSet (:62-93), LagrangePolynomialSet (:96-122)."""
import numpy

from . import polynomial_set, reference_element, runtime


def get_lagrange_points(nodes):
    """The single evaluation point of each point-evaluation node."""
    points = []
    for node in nodes:
        pt, = node.get_point_dict()
        points.append(pt)
    return points


def make_dmat(x):
    """Differentiation matrix (acting on basis-function values) and barycentric weights."""
    x = numpy.asarray(x, dtype=float).reshape(-1)
    diff = x[None, :] - x[:, None]
    numpy.fill_diagonal(diff, 1.0)
    wts = 1.0 / numpy.prod(diff, axis=0)
    dmat = (wts[:, None] / wts[None, :]) / diff
    numpy.fill_diagonal(dmat, dmat.diagonal() - dmat.sum(axis=0))
    return dmat, wts


class LagrangeLineExpansionSet:
    """Lagrange polynomials on given points of an interval."""

    def __init__(self, ref_el, pts):
        if ref_el.get_shape() != reference_element.LINE:
            raise ValueError("Must have a line")
        self.ref_el = ref_el
        self.points = pts
        self.x = numpy.array(pts, dtype="d").flatten()
        self.degree = len(self.x) - 1
        self.recurrence_order = self.degree + 1
        self.variant = None
        self.continuity = None
        self.scale = 1.0
        self._dev = None

    def device_line(self):
        if self._dev is None:
            self._dev = runtime.LineLagrange(self.x)
        return self._dev

    def get_num_members(self, n):
        return len(self.points)

    def get_points(self):
        return self.points

    def get_scale(self, n, cell=0):
        return self.scale

    def get_dmats(self, degree, cell=0):
        return [make_dmat(self.x)[0].T]

    def _tabulate(self, n, pts, order=0):
        pts = numpy.asarray(pts, dtype=float)
        single = pts.ndim == 1 and pts.shape[0] == 1
        out = self.device_line().tabulate_batch(order, pts.reshape(1, -1)).cpu().numpy()[0]
        result = {(r,): numpy.ascontiguousarray(out[r]) for r in range(order + 1)}
        if single:
            result = {a: v[..., 0] for a, v in result.items()}
        return result

    def tabulate(self, n, pts):
        if len(pts) == 0:
            return numpy.array([])
        return self._tabulate(n, pts)[(0,)]


class LagrangePolynomialSet(polynomial_set.PolynomialSet):
    def __init__(self, ref_el, pts, shape=()):
        if ref_el.get_shape() != reference_element.LINE:
            raise ValueError("Invalid reference element type.")
        es = LagrangeLineExpansionSet(ref_el, pts)
        nexp = es.get_num_members(es.degree)
        ncomp = int(numpy.prod(shape, dtype=int))
        if shape == ():
            coeffs = numpy.eye(nexp, dtype="d")
        else:
            coeffs = numpy.zeros((ncomp * nexp, *shape, nexp), "d")
            for c, idx in enumerate(numpy.ndindex(shape)):
                coeffs[(range(c * nexp, (c + 1) * nexp), *idx, range(nexp))] = 1.0
        super().__init__(ref_el, es.degree, es.degree, es, coeffs)

    def device_polyset(self):
        """Batched device form with the interface of runtime.SimplexPolySet (points (nreq, npts, 1))."""
        if self.coeffs.ndim != 2 or not numpy.array_equal(self.coeffs, numpy.eye(self.coeffs.shape[0])):
            raise NotImplementedError("batched tabulation of 1-D Lagrange sets with non-identity coefficients")
        return _LineBatch(self.expansion_set.device_line())

    def tabulate(self, pts, jet_order=0):
        base = self.expansion_set._tabulate(self.embedded_degree, pts, jet_order)
        # the nodal coefficients over the primal Lagrange basis are the identity
        # (barycentric values at the nodes are exact Kronecker deltas)
        if self.coeffs.ndim == 2 and numpy.array_equal(self.coeffs, numpy.eye(self.coeffs.shape[0])):
            return base
        return {a: numpy.tensordot(self.coeffs, v, axes=(-1, 0)) for a, v in base.items()}


class _LineBatch:
    """fx_line_tabulate_batch behind the calling convention of the simplex polynomial sets."""
    sd, vdim, value_shape = 1, 1, ()

    def __init__(self, line):
        self.line = line
        self.ndof = line.nn
        self.ctx = line.ctx

    def out_shape(self, order, nreq, npts):
        return (nreq, order + 1, self.ndof, npts)

    def tabulate_batch(self, order, pts, verts=None, out=None, stream=None, mapping=None):
        if verts is not None or mapping not in (None, "affine"):
            raise NotImplementedError("per-request cells for 1-D Lagrange sets")
        pts = runtime._as_device(pts, self.ctx)
        if pts.dim() != 3 or pts.shape[2] != 1:
            raise ValueError(f"points must have shape (nreq, npts, 1), got {tuple(pts.shape)}")
        return self.line.tabulate_batch(order, pts.reshape(pts.shape[0], pts.shape[1]), out=out, stream=stream)
