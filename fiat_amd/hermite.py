"""Cubic Hermite elements on simplices: P3 with the value and the gradient at every vertex and the value
at the barycentre of every triangular face.  The gradient dofs are derivative functionals -- their
Vandermonde rows come from the order-1 expansion tables (dual_set.to_riesz; FIAT/dual_set.py:175-205).
Behaviour as FIAT/hermite.py:11-80."""
from . import finite_element, functional, polynomial_set
from .dof_layout import DofLayout


def hermite_dofs(cell):
    lay = DofLayout(cell)
    sd = lay.sd
    axes = [tuple(int(i == j) for i in range(sd)) for j in range(sd)]
    for v in lay.entities(0):
        x = cell.get_vertices()[v]
        lay.place(0, v, [functional.PointEvaluation(cell, x)] + [functional.PointDerivative(cell, x, a) for a in axes])
    if sd >= 2:
        lay.lattice(2, 3, lambda _, pts: [functional.PointEvaluation(cell, x) for x in pts])
    return lay.dual_set()


class CubicHermite(finite_element.CiarletElement):
    def __init__(self, ref_el, deg=3):
        assert deg == 3
        super().__init__(polynomial_set.ONPolynomialSet(ref_el, 3), hermite_dofs(ref_el), 3)
