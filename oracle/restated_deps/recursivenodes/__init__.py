"""TEST INFRASTRUCTURE ONLY -- restatement of the third-party package
``recursivenodes`` (T. Isaac, "Recursive, parameter-free, explicitly defined
interpolation nodes for simplices", SISC 42(6), 2020), which the reference
lists unpinned in pyproject.toml:13 and which is absent from this image.

Only the surface the reference's hot path touches is restated
(FIAT/reference_element.py:30,89-98 and FIAT/quadrature.py:13,102,123,179).
It exists so that the *unmodified* reference can be imported in the build
container to generate golden vectors (tests/golden/make_golden.py).  It is
never imported by the product (fiat_amd/) and never needed on the GPU box.
"""
