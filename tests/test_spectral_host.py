"""Spectral point families on the CPU: the lattices of fiat_amd.reference_element.make_lattice against closed forms
(1-D: the families are defined mathematically) and against fixtures produced by the reference
(tests/golden/make_golden_spectral.py; on simplices *given* our restatement of the absent ``recursivenodes``
package -- parity unpinned with respect to that package itself)."""
import math

import numpy as np
import pytest

from fiat_amd import reference_element as re_


def test_gll_and_gl_closed_forms():
    I = re_.ufc_simplex(1).get_vertices()
    x = 2.0 * np.array(re_.make_lattice(I, 4, variant="gll"))[:, 0] - 1.0
    assert np.allclose(x, [-1.0, -math.sqrt(3.0 / 7.0), 0.0, math.sqrt(3.0 / 7.0), 1.0], atol=1e-15)
    x = 2.0 * np.array(re_.make_lattice(I, 3, variant="gll"))[:, 0] - 1.0
    assert np.allclose(x, [-1.0, -1.0 / math.sqrt(5.0), 1.0 / math.sqrt(5.0), 1.0], atol=1e-15)
    x = 2.0 * np.array(re_.make_lattice(I, 2, variant="gl"))[:, 0] - 1.0
    assert np.allclose(x, [-math.sqrt(0.6), 0.0, math.sqrt(0.6)], atol=1e-15)
    x = 2.0 * np.array(re_.make_lattice(I, 2, variant="lgc"))[:, 0] - 1.0
    assert np.allclose(x, [-1.0, 0.0, 1.0], atol=1e-15)
    with pytest.raises(ValueError):
        re_.make_lattice(I, 2, variant="nope")


@pytest.mark.parametrize("sd", [1, 2, 3])
@pytest.mark.parametrize("variant", ["gll", "gl", "lgc", "gc", "equispaced_interior"])
def test_lattices_match_reference_code_path(golden, sd, variant):
    G = golden("spectral")
    V = re_.ufc_simplex(sd).get_vertices()
    for n in (1, 2, 3, 4, 5):
        got = np.array(re_.make_lattice(V, n, variant=variant))
        assert np.max(np.abs(got - G[f"lattice/{variant}/sd{sd}/n{n}"])) < 1e-14
    got = np.array(re_.make_lattice(V, 4, 1, variant=variant)).reshape(-1, sd)
    assert np.max(np.abs(got - G[f"lattice/{variant}/sd{sd}/n4_int1"]), initial=0.0) < 1e-14


def test_symmetry_of_recursive_lattices():
    """The recursive rule is symmetric: permuting the vertices permutes the lattice (Isaac 2020, section 3)."""
    V = np.array(re_.ufc_simplex(2).get_vertices())
    pts = np.array(re_.make_lattice(V, 4, variant="gll"))
    bary = np.column_stack([1.0 - pts.sum(1), pts])
    ref = {tuple(np.round(b, 12)) for b in bary}
    for perm in ([1, 0, 2], [2, 1, 0], [1, 2, 0]):
        assert {tuple(np.round(b[perm], 12)) for b in bary} == ref
