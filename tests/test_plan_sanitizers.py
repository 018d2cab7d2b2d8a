"""The host planning code of the C ABI (fiat_amd/csrc/plan.hpp: recurrence programs, C0 transform, cooperative
schedule, fragment packings) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and run on the CPU
(GPU sanitizers are not available on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_plan_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "plan_sanitize"
    src = os.path.join(ROOT, "tests", "native", "plan_sanitize.cpp")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-o", str(exe), src], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "programs ok" in run.stdout
