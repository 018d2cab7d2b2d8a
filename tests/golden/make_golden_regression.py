"""The reference's REGRESSION suite (test/FIAT/regression/test_regression.py) keeps its expected values in JSON files of a
separate repository that cannot be fetched here; the suite's own fallback is to regenerate them with its ``create_data``
recipes.  This script runs those recipes on the *unmodified reference* in the build container and stores the numbers:

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 python -B tests/golden/make_golden_regression.py

* ``dmats_tet3`` / ``dmats_line3``: ONPolynomialSet(default tetrahedron / line, 3).get_dmats()          (:81-114)
* ``exp_tri3_phi`` / ``_dphi``: ExpansionSet(default triangle).tabulate / tabulate_derivatives, lattice 3  (:117-146)
* ``jet_tet``: TetrahedronExpansionSet.tabulate_jet(1, lattice 2, order 2)                               (:149-166)
* ``quad_<family>_<dim>_<degree>_<alpha>``: element.tabulate(3, make_quadrature(simplex, 3) points) for the (family, dim,
  degree) rows of the parametrisation (:200-272) whose family is in scope; "point" variant for Regge / HHJ
  as there (:283-284);
* ``ptcell_*`` (not from the regression suite): Raviart-Thomas on the interval (degrees 1-3, three variants) with first
  derivatives, and DG of degree 0 on a point -- the elements of test_fiat.py's nodality list that need polynomials on a POINT cell.
Plain numbers only."""
import os

import numpy as np

from FIAT import expansions, make_quadrature, polynomial_set, reference_element, supported_elements, ufc_simplex

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ([("Lagrange", d, k) for d in (1, 2, 3) for k in (1, 2, 3)] + [("Discontinuous Lagrange", d, k) for d in (1, 2, 3) for k in (0, 1, 2)]
         + [(f, d, k) for f in ("Brezzi-Douglas-Marini", "Raviart-Thomas", "Nedelec 1st kind H(curl)", "Nedelec 2nd kind H(curl)")
            for d in (2, 3) for k in (1, 2, 3)]
         + [("Regge", d, k) for d in (2, 3) for k in (0, 1, 2)] + [("Hellan-Herrmann-Johnson", 2, k) for k in (0, 1, 2)])


def main():
    out = {}
    for name, cell in (("dmats_tet3", reference_element.DefaultTetrahedron()), ("dmats_line3", reference_element.DefaultLine())):
        out[name] = np.array(polynomial_set.ONPolynomialSet(ref_el=cell, degree=3).get_dmats())
    E = reference_element.DefaultTriangle()
    pts = reference_element.make_lattice(E.get_vertices(), 3)
    Phis = expansions.ExpansionSet(E)
    out["exp_tri3_pts"] = np.array(pts)
    out["exp_tri3_phi"] = np.array(Phis.tabulate(3, pts))
    d = Phis.tabulate_derivatives(3, pts)
    out["exp_tri3_dphi_value"] = np.array([[p[0] for p in row] for row in d])
    out["exp_tri3_dphi_grad"] = np.array([[p[1] for p in row] for row in d])
    T = reference_element.DefaultTetrahedron()
    jpts = reference_element.make_lattice(T.get_vertices(), 2)
    jet = expansions.TetrahedronExpansionSet(T).tabulate_jet(1, jpts, 2)
    out["jet_tet_pts"] = np.array(jpts)
    for r, datum in enumerate(jet):
        out[f"jet_tet_{r}"] = np.array(datum)
    for family, dim, degree in CASES:
        kwargs = {"variant": "point"} if family in {"Regge", "Hellan-Herrmann-Johnson"} else {}
        domain = ufc_simplex(dim)
        element = supported_elements[family](domain, degree, **kwargs)
        points = make_quadrature(domain, 3).get_points()
        table = element.tabulate(3, points)
        key = f"quad_{family.replace(' ', '_')}_{dim}_{degree}"
        out[key + "_pts"] = np.array(points)
        for alpha, v in table.items():
            out[key + "_" + "".join(map(str, alpha))] = np.asarray(v)
    # (not part of the regression suite: the elements over POINT cells of test_fiat.py's nodality list, tabulated)
    import FIAT
    I = ufc_simplex(1)
    ipts = np.array([[0.0], [0.3], [0.5], [0.77], [1.0]])
    out["ptcell_pts"] = ipts
    for k in (1, 2, 3):
        for variant in ("integral", "integral(1)", "point"):
            el = FIAT.RaviartThomas(I, k, variant=variant)
            tab = el.tabulate(1, ipts)
            out[f"ptcell_rt{k}_{variant}_0"], out[f"ptcell_rt{k}_{variant}_1"] = np.asarray(tab[(0,)]), np.asarray(tab[(1,)])
    dg = FIAT.DiscontinuousLagrange(ufc_simplex(0), 0)
    out["ptcell_dg0"] = np.asarray(dg.tabulate(0, [()])[()])
    path = os.path.join(HERE, "regression.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
