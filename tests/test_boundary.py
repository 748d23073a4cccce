"""The drop-in boundary outside Python: include/vidc_faiss_adapter.h compiled against the Faiss interface shim
(tests/faiss_shim: Faiss itself is not in this image) and the raw C-ABI from a plain C program.

CPU (no GPU needed): both programs compile and link against the built libvidc.so -- the header is valid C, the adapter
is valid C++ against the InvertedLists / nsg::Graph / IndexIVF interfaces.  GPU: they run; adapter_smoke drives the seven
container classes and search_IVF_defer_id_decoding the way test_compressed_ivfs.py / test_altid.py do, c_abi_smoke checks
KAT1 (head + words of the reference's own codec.cpp) through nothing but vidc.h."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vector_db_id_compression_amd")


def _build(tmp_path, which):
    from vector_db_id_compression_amd import build

    build.build()
    exe = str(tmp_path / which)
    if which == "adapter_smoke":
        cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O1", "-fopenmp", "-Wall", "-I", os.path.join(ROOT, "tests", "faiss_shim"),
               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "adapter_smoke.cpp")]
    else:
        cmd = [shutil.which("gcc") or "gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
               os.path.join(ROOT, "tests", "c_abi_smoke.c")]
    cmd += ["-L", PKG, "-lvidc", "-Wl,-rpath," + PKG, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("which", ["adapter_smoke", "c_abi_smoke"])
def test_compiles_and_links(tmp_path, which):
    assert os.path.exists(_build(tmp_path, which))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["adapter_smoke", "c_abi_smoke"])
def test_runs_on_the_gpu(tmp_path, which):
    exe = _build(tmp_path, which)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("adapter smoke ok" if which == "adapter_smoke" else "c abi smoke ok") in r.stdout
