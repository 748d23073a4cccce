"""The host's passes over the CSR offsets (csrc/host_passes.h): AVX2 forms against the scalar loops, built with g++ (no HIP, no GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("no_avx2", [False, True])
def test_host_passes_match_their_scalar_forms(tmp_path, no_avx2):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not found")
    exe = str(tmp_path / "host_passes_test")
    subprocess.check_call([gxx, "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "host_passes_test.cpp")])
    env = dict(os.environ)
    if no_avx2:
        env["VIDC_NO_AVX2"] = "1"  # the dispatchers fall back to the scalar loops
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "host passes ok" in out.stdout, out.stdout + out.stderr
