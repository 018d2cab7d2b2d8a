"""PolynomialSet facade: polynomials as coefficient tensors over an expansion set;
``tabulate`` is the coeffs x expansion-values contraction, done by the HIP kernel
(recurrence and contraction fused, f64 MFMA).

Mirrors FIAT/polynomial_set.py: PolynomialSet (:42-107), ONPolynomialSet
(:110-134), spanning_basis (:160-168), polynomial_set_union_normalized (:171-187),
construct_new_coeffs (:190-217)."""
import numpy

from . import expansions, runtime
from .polynomial_set_util import mis  # noqa: F401  (re-exported, FIAT.polynomial_set.mis)


class PolynomialSet:
    """coeffs[i, *value_shape, k]: member i = sum_k coeffs[i, :, k] phi_k."""

    def __init__(self, ref_el, degree, embedded_degree, expansion_set, coeffs):
        self.ref_el = ref_el
        self.num_members = coeffs.shape[0]
        self.degree = degree
        self.embedded_degree = embedded_degree
        self.expansion_set = expansion_set
        self.coeffs = coeffs
        self._dev = None

    def device_polyset(self):
        """The device-resident form used by the batched API (created once)."""
        if self._dev is None:
            self._dev = self.expansion_set.device_polyset(self.embedded_degree, coeffs=self.coeffs,
                                                          value_shape=self.get_shape())
        return self._dev

    def tabulate(self, pts, jet_order=0):
        """{alpha: (num_members, *value_shape, npts)} for all |alpha| <= jet_order."""
        sd = self.ref_el.get_spatial_dimension()
        if sd == 0:     # point cell: coefficients times the constant 1 (host arithmetic, FIAT/expansions.py:638-649)
            base = self.expansion_set._tabulate(self.embedded_degree, pts, jet_order)[()]
            return {(): numpy.dot(self.coeffs, base)}
        pts = numpy.asarray(pts, dtype=float)
        single = pts.ndim == 1
        out = runtime.fetch(self.device_polyset().tabulate_batch(jet_order, pts.reshape(1, -1, sd)))[0]
        keys = [a for k in range(jet_order + 1) for a in mis(sd, k)]
        result = {a: numpy.ascontiguousarray(out[t]) for t, a in enumerate(keys)}
        if single:
            result = {a: v[..., 0] for a, v in result.items()}
        return result

    def tabulate_new(self, pts):
        sd = self.ref_el.get_spatial_dimension()
        return self.tabulate(pts)[(0,) * sd]

    def get_expansion_set(self):
        return self.expansion_set

    def get_dmats(self, cell=0):
        """Differentiation matrices of the expansion set at this set's degree (FIAT/polynomial_set.py:91-92)."""
        return self.expansion_set.get_dmats(self.degree, cell=cell)

    def get_coeffs(self):
        return self.coeffs

    def get_num_members(self):
        return self.num_members

    def get_degree(self):
        return self.degree

    def get_embedded_degree(self):
        return self.embedded_degree

    def get_reference_element(self):
        return self.ref_el

    def get_shape(self):
        return self.coeffs.shape[1:-1]

    def take(self, items):
        return PolynomialSet(self.ref_el, self.degree, self.embedded_degree, self.expansion_set,
                             numpy.take(self.coeffs, items, 0))

    def __len__(self):
        return self.num_members


class ONPolynomialSet(PolynomialSet):
    """Identity coefficients over the expansion set, repeated per value component."""

    def __init__(self, ref_el, degree, shape=(), **kwargs):
        es = expansions.ExpansionSet(ref_el, **kwargs)
        ncomp = int(numpy.prod(shape, dtype=int))
        nexp = es.get_num_members(degree)
        if shape == ():
            coeffs = numpy.eye(nexp)
        else:
            coeffs = numpy.zeros((ncomp * nexp, *shape, nexp))
            for c, idx in enumerate(numpy.ndindex(shape)):
                coeffs[(range(c * nexp, (c + 1) * nexp), *idx, range(nexp))] = 1.0
        super().__init__(ref_el, degree, degree, es, coeffs)


class ONSymTensorPolynomialSet(PolynomialSet):
    """Symmetric-matrix-valued polynomials (FIAT/polynomial_set.py:220-249): for every component pair i <= j one
    copy of the expansion set with the value e_i e_j^T + e_j e_i^T (the diagonal ones: e_i e_i^T)."""

    def __init__(self, ref_el, degree, size=None, **kwargs):
        es = expansions.ExpansionSet(ref_el, **kwargs)
        if size is None:
            size = ref_el.get_spatial_dimension()
        nexp = es.get_num_members(degree)
        pairs = [(i, j) for i in range(size) for j in range(i, size)]
        coeffs = numpy.zeros((len(pairs) * nexp, size, size, nexp))
        members = numpy.arange(nexp)
        for block, (i, j) in enumerate(pairs):
            coeffs[block * nexp + members, i, j, members] = 1.0
            coeffs[block * nexp + members, j, i, members] = 1.0
        super().__init__(ref_el, degree, degree, es, coeffs)


class TracelessTensorPolynomialSet(PolynomialSet):
    """Trace-free matrix-valued polynomials (FIAT/polynomial_set.py:252-282): one copy of the expansion set for every
    component (i, j) except the last diagonal one; a diagonal copy carries e_i e_i^T - e_n e_n^T, which keeps the trace
    at zero.  size^2 - 1 blocks of nexp members, blocks in row-major component order -- the space of the GLS elements."""

    def __init__(self, ref_el, degree, size=None, **kwargs):
        es = expansions.ExpansionSet(ref_el, **kwargs)
        if size is None:
            size = ref_el.get_spatial_dimension()
        nexp = es.get_num_members(degree)
        blocks = [(i, j) for i in range(size) for j in range(size)][:-1]
        coeffs = numpy.zeros((len(blocks) * nexp, size, size, nexp))
        members = numpy.arange(nexp)
        for b, (i, j) in enumerate(blocks):
            coeffs[b * nexp + members, i, j, members] = 1.0
            if i == j:
                coeffs[b * nexp + members, size - 1, size - 1, members] = -1.0
        super().__init__(ref_el, degree, degree, es, coeffs)


def spanning_basis(A, nullspace=False, rtol=1e-10):
    """Orthonormal basis of the row space (or its complement) of A via SVD."""
    Aflat = A.reshape(A.shape[0], -1)
    _, sig, vt = numpy.linalg.svd(Aflat, full_matrices=True)
    atol = rtol * (sig[0] + 1)
    num_sv = int(numpy.count_nonzero(numpy.abs(sig) > atol))
    basis = vt[num_sv:] if nullspace else vt[:num_sv]
    return numpy.reshape(basis, (-1, *A.shape[1:]))


def construct_new_coeffs(ref_el, A, B):
    """Stack the coefficients of two sets, zero-extending the lower-degree one."""
    if A.get_expansion_set().continuity != B.get_expansion_set().continuity:
        raise ValueError("Continuity of expansion sets does not match.")
    da, db = A.get_embedded_degree(), B.get_embedded_degree()
    if da == db:
        return numpy.concatenate((A.coeffs, B.coeffs), axis=0)
    if A.get_expansion_set().continuity is not None:
        raise NotImplementedError("Extending of coefficients is not implemented for PolynomialSets "
                                  "with continuity and different degrees")
    higher, lower = (A, B) if da > db else (B, A)
    diff = higher.coeffs.shape[-1] - lower.coeffs.shape[-1]
    pad = [(0, 0)] * (lower.coeffs.ndim - 1) + [(0, diff)]
    return numpy.concatenate((numpy.pad(lower.coeffs, pad), higher.coeffs), axis=0)


def polynomial_set_union_normalized(A, B):
    """A set spanning span(A) + span(B) (orthonormal coefficient rows)."""
    assert A.get_reference_element() == B.get_reference_element()
    coeffs = spanning_basis(construct_new_coeffs(A.get_reference_element(), A, B))
    return PolynomialSet(A.get_reference_element(), max(A.get_degree(), B.get_degree()),
                         max(A.get_embedded_degree(), B.get_embedded_degree()),
                         A.get_expansion_set(), coeffs)


def make_bubbles(ref_el, degree, codim=0, shape=(), scale="L2 piola"):
    """The members of the C0 ("bubble") hierarchy of degree <= ``degree`` that belong to the entities of co-dimension ``codim``
    -- for codim 0 the interior bubbles of the cell -- as a polynomial set, every value component of ``shape`` carrying its own
    copy (FIAT/polynomial_set.py:285-301; test/FIAT/unit/test_polynomial.py:123-135 checks their duality with the "dual" variant)."""
    full = ONPolynomialSet(ref_el, degree, shape=shape, scale=scale, variant="bubble")
    sd = ref_el.get_spatial_dimension()
    if sd == 0:
        return full
    owners = expansions.polynomial_entity_ids(ref_el, degree, continuity="C0")[sd - codim]
    members = [i for entity in sorted(owners) for i in owners[entity]]
    if shape != ():
        per_component = full.get_num_members() // int(numpy.prod(shape, dtype=int))
        members = [i + c * per_component for i in members for c in range(int(numpy.prod(shape, dtype=int)))]
    return full.take(members)
