"""Runs the C++ plug-in surface tests (tests/cpp/test_host_plugin.cpp): the lduMatrix::solver
classes GKOCG / GKOBiCGStab / GKOGMRES of ogl_amd/host/OGLAdapter.H over the MiniFoam stand-in."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_plugin")


def _run(which):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    p = subprocess.run([EXE, which], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 failure(s)" in p.stdout
    return p.stdout


def test_cpp_plugin_surface_cpu():
    out = _run("cpu")
    assert out.count("[  OK  ]") >= 6


@pytest.mark.gpu
def test_cpp_plugin_surface_gpu():
    out = _run("gpu")
    assert out.count("[  OK  ]") >= 3


def test_host_logic_under_asan_ubsan():
    """host_matrix.cpp + common.cpp (pure host, no HIP) built with g++ -fsanitize=address,undefined and
    driven through the ogl_host_* ABI (tests/cpp/asan_host_logic.cpp)."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp, "asan_host_logic"], stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(cpp, "asan_host_logic")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "0 failure(s)" in p.stdout and "ERROR" not in p.stderr


@pytest.mark.parametrize("flags", [[], ["-DWITH_ESI_VERSION"]], ids=["Foundation", "ESI"])
def test_openfoam_translation_unit_parses_against_the_api_stub(flags):
    """ogl_amd/foam/GKOSolvers.C is compiled in an OpenFOAM tree, which this image does not have.  A header-only
    stub of the OpenFOAM declarations it uses (tests/cpp/foam_stub, on top of MiniFoam.H) lets `g++ -fsyntax-only`
    catch typos and signature slips in the adapter translation unit before a maintainer's wmake does."""
    p = subprocess.run(["g++", "-std=c++17", "-Wall", "-fsyntax-only", *flags,
                        "-I", os.path.join(ROOT, "tests", "cpp", "foam_stub"), "-I", os.path.join(ROOT, "ogl_amd", "host"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "ogl_amd", "foam", "GKOSolvers.C")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "error" not in p.stderr, p.stderr[-3000:]
