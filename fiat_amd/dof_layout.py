"""Degrees of freedom described block by block, entity by entity.

Every in-scope family is "a polynomial space + a list of dof blocks"; a block says, for all sub-entities of
one dimension, which functionals live there:

* ``lattice``   functionals made from the points of the entity's lattice (point values, tangential /
                normal components at points, ...),
* ``moments``   integrals against (scalar test polynomial on the entity) x (a fixed tensor of the entity:
                a tangent, the scaled normal, t t^T, n n^T, a Cartesian axis ...),
* ``field_moments``  integrals against a vector field given on the REFERENCE entity and pulled to the
                entity covariantly or contravariantly (the interior dofs of BDM, all dofs of N2curl),
* ``place``     anything else (vertex jets, normal-derivative averages).

The family modules (nedelec.py, raviart_thomas.py, regge.py ...) are short tables over these blocks; this
module has no counterpart in the reference -- there each family spells its loops out by hand
(FIAT/nedelec.py:99-178, raviart_thomas.py:56-122, regge.py:17-60 define the same functionals).  The
functionals themselves are data (functional.py); their arithmetic runs on the device in DualSet.to_riesz."""
import numpy

from . import functional, polynomial_set
from .check_format_variant import parse_quadrature_scheme
from .dual_set import DualSet
from .quadrature import FacetQuadratureRule


class DofLayout:
    def __init__(self, cell):
        self.cell = cell
        self.sd = cell.get_spatial_dimension()
        self.topology = cell.get_topology()
        self.functionals = []
        self.owned = {dim: {e: [] for e in sorted(self.topology[dim])} for dim in sorted(self.topology)}

    # -- bookkeeping -------------------------------------------------------------------------------
    def entities(self, dim):
        return sorted(self.topology[dim])

    def place(self, dim, entity, new):
        """Append functionals owned by sub-entity (dim, entity); they are numbered in order of arrival."""
        new = list(new)
        start = len(self.functionals)
        self.functionals.extend(new)
        self.owned[dim][entity] = self.owned[dim][entity] + list(range(start, start + len(new)))

    def dual_set(self, cls=DualSet):
        return cls(self.functionals, self.cell, self.owned)

    def parts(self):
        return self.functionals, self.cell, self.owned

    # -- blocks ------------------------------------------------------------------------------------
    def lattice(self, dim, lattice_order, make, variant=None, owner=None):
        """``make(entity, points) -> functionals`` for the interior lattice points of every entity of
        dimension ``dim``; ``owner = (dim, entity)`` hands all of them to one entity (broken spaces)."""
        for entity in self.entities(dim):
            points = self.cell.make_points(dim, entity, lattice_order, variant=variant)
            self.place(*(owner or (dim, entity)), make(entity, points))

    def _reference_rule(self, dim, quad_degree, scheme):
        return parse_quadrature_scheme(self.cell.construct_subelement(dim), quad_degree, scheme)

    def moments(self, dim, test_degree, quad_degree, frames, *, scheme=None, average=True, tests=None,
                frame_major=False, tag=None):
        """Moments against p_m(x) F_c over every entity E of dimension ``dim``: ``frames(E)`` is the list of
        fixed tensors F_c (each of the element's value shape), p_m runs over an orthonormal basis of
        P_test_degree on the reference entity (or ``tests(rule) -> (nm, nq)``).  Numbering: p outer, F inner,
        or the other way round with ``frame_major``.  ``average``: the rule keeps the reference weights."""
        if test_degree < 0:
            return
        rule = self._reference_rule(dim, quad_degree, scheme)
        if tests is None:
            on = polynomial_set.ONPolynomialSet(self.cell.construct_subelement(dim), test_degree)
            scalars = on.tabulate(rule.get_points())[(0,) * dim]
        else:
            scalars = numpy.asarray(tests(rule), dtype=float)
        for entity in self.entities(dim):
            Q = FacetQuadratureRule(self.cell, dim, entity, rule, avg=average)
            tensors = [numpy.asarray(F, dtype=float) for F in frames(entity)]
            pairs = ([(F, p) for F in tensors for p in scalars] if frame_major
                     else [(F, p) for p in scalars for F in tensors])
            self.place(dim, entity, (functional.FrobeniusIntegralMoment(self.cell, Q, F[..., None] * p, tag)
                                     for F, p in pairs))

    def component_moments(self, dim, test_degree, quad_degree, *, scheme=None):
        """Moments of every Cartesian component against an orthonormal basis of P_test_degree on the
        entities of dimension ``dim`` (component outer, test function inner)."""
        if test_degree < 0:
            return
        rule = self._reference_rule(dim, quad_degree, scheme)
        on = polynomial_set.ONPolynomialSet(self.cell.construct_subelement(dim), test_degree)
        scalars = on.tabulate(rule.get_points())[(0,) * dim]
        for entity in self.entities(dim):
            Q = FacetQuadratureRule(self.cell, dim, entity, rule)
            self.place(dim, entity, (functional.IntegralMoment(self.cell, Q, p, (c,), (self.sd,))
                                     for c in range(self.sd) for p in scalars))

    def field_moments(self, dim, fields, quad_degree, pullback, *, scheme=None):
        """Moments against vector fields known on the reference entity: ``fields(points) -> (nf, dim, nq)``.
        ``pullback`` = "covariant": the field J^-T f, "contravariant": J f / |J| with J the Jacobian of
        reference entity -> entity."""
        rule = self._reference_rule(dim, quad_degree, scheme)
        reference_fields = numpy.asarray(fields(rule.get_points()), dtype=float)
        for entity in self.entities(dim):
            Q = FacetQuadratureRule(self.cell, dim, entity, rule)
            J = Q.jacobian()
            if pullback == "covariant":
                M = numpy.linalg.pinv(J).T
            elif pullback == "contravariant":
                M = J / Q.jacobian_determinant()
            else:
                raise ValueError(f"unknown pullback {pullback!r}")
            mapped = numpy.einsum("ab,fbq->faq", M, reference_fields)
            self.place(dim, entity, (functional.FrobeniusIntegralMoment(self.cell, Q, f) for f in mapped))


def augmented_vector_space(cell, k, lift):
    """The polynomial space  P_k^d  +  { lift(p, x) : p of exact degree k }  as a PolynomialSet over the
    vector-valued orthonormal basis of degree k + 1 (H(div): lift = p x; H(curl): p rot x, p e_i x x).

    The lifted fields are L2-projected onto (P_{k+1})^d at a rule exact for degree 2k + 2 (tabulated on the
    device).  Their components along P_k^d are already in the space, so only the degree-(k+1) part of each
    projection is kept and an orthonormal basis of those parts (SVD) is appended to the unit coefficient rows
    of P_k^d.  The span equals the one FIAT builds (FIAT/raviart_thomas.py:17-53, nedelec.py:17-96); the
    nodal basis of the Ciarlet element does not depend on the basis chosen for it."""
    from . import expansions
    from .quadrature import create_quadrature
    sd = cell.get_spatial_dimension()
    below, same, above = (expansions.polynomial_dimension(cell, j) if j >= 0 else 0 for j in (k - 1, k, k + 1))
    scalar = polynomial_set.ONPolynomialSet(cell, k + 1)
    rule = create_quadrature(cell, 2 * (k + 1))
    x, w = rule.get_points(), rule.get_weights()
    phi = scalar.tabulate(x)[(0,) * sd]                         # (above, nq)
    fields = numpy.asarray(lift(phi[below:same], x.T))          # (nf, sd, nq)
    proj = numpy.einsum("fcq,q,jq->fcj", fields, w, phi)
    proj[:, :, :same] = 0.0
    _, sing, vt = numpy.linalg.svd(proj.reshape(len(proj), -1), full_matrices=False)
    rank = int(numpy.count_nonzero(sing > 1e-10 * max(sing[0], 1.0)))
    coeffs = numpy.zeros((sd * same + rank, sd, above))
    for c in range(sd):
        coeffs[c * same + numpy.arange(same), c, numpy.arange(same)] = 1.0
    coeffs[sd * same:] = vt[:rank].reshape(rank, sd, above)
    return polynomial_set.PolynomialSet(cell, k + 1, k + 1, scalar.get_expansion_set(), coeffs)
