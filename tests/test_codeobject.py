"""Build check on the device code inside libfiat_amd.so (no GPU needed): gfx950 only, the benchmark kernels free of
scratch, and a scratch budget for everything else.  Background: the one unexplained GPU fault of round 1 came from a
stacked-kernel instance with 512 registers, 90+ SGPR spills and 516 B of scratch per lane that was never registered
(DESIGN.md 4.5b); instances are now admitted only within this budget, checked here from the code-object metadata
(tools/codeobject_report.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SCRATCH_BUDGET = 128      # bytes per lane; today's maximum: tabulate_simplex_stacked<3, 6, 2, 1, 0, 1, false, 4> with 104
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler"),
                                reason="needs the LLVM tools of ROCm")


@pytest.fixture(scope="module")
def report():
    import codeobject_report
    return codeobject_report.kernels()


def test_only_gfx950_code(report):
    kernels, targets = report
    assert sorted(targets) == ["hipv4-amdgcn-amd-amdhsa--gfx950", "host-x86_64-unknown-linux-gnu-"]
    assert len(kernels) > 300


def test_benchmark_kernels_have_no_scratch(report):
    kernels, _ = report
    by_prefix = lambda p: [k for k in kernels if k["name"].startswith(p)]   # noqa: E731
    # paired kernel: P3 tetrahedron (UNIFORM instance = the headline), RT2 and N2 incl. their Piola instances
    heads = [k for k in by_prefix("_ZN3fxk21tabulate_simplex_pairILi3ELi3ELi1ELi20ELi6ELi8ELb1E")]
    heads += by_prefix("_ZN3fxk21tabulate_simplex_pairILi3ELi2ELi1ELi45E") + by_prefix("_ZN3fxk21tabulate_simplex_pairILi3ELi2ELi1ELi60E")
    heads += [k for k in by_prefix("_ZN3fxk24tabulate_simplex_stackedILi3ELi6ELi3ELi2ELi0E")]      # DG P6 with Hessians (C4)
    assert len(heads) >= 8
    for k in heads:
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, k


def test_scratch_budget(report):
    kernels, _ = report
    worst = max(kernels, key=lambda k: k["scratch"])
    assert worst["scratch"] <= SCRATCH_BUDGET, worst
    # ... and few kernels use any: 17 of 583 at the end of round 3 (was 9 of 332), all <= 96 B
    assert sum(1 for k in kernels if k["scratch"]) <= max(16, len(kernels) // 30)
