"""Runs the C++ plug-in surface tests (tests/cpp/test_host_plugin.cpp): the lduMatrix::solver
classes GKOCG / GKOBiCGStab / GKOGMRES of ogl_amd/host/OGLAdapter.H over the MiniFoam stand-in (tests/cpp/MiniFoam.H: test infrastructure)."""
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
                        "-I", os.path.join(ROOT, "tests", "cpp", "foam_stub"), "-I", os.path.join(ROOT, "tests", "cpp"),
                        "-I", os.path.join(ROOT, "ogl_amd", "host"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "ogl_amd", "foam", "GKOSolvers.C")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "error" not in p.stderr, p.stderr[-3000:]


def test_host_layout_builders_under_sanitizers(tmp_path):
    """The pure-host part of the library (pattern -> per-chunk half storage, compressed chunked ELL; the code the GPU
    AddressSanitizer of this pool cannot reach) built with -fsanitize=address,undefined and run on one pattern of every
    class: multi-block meshes with even and odd line lengths, random bands with an odd row count, a box, one cell, an
    octree and a Voronoi mesh.  Any report makes the run exit non-zero."""
    import numpy as np
    from ogl_amd import capi, synthetic
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ogl_amd", "csrc"),
           os.path.join(ROOT, "tools", "host_sanitize.cpp"), os.path.join(ROOT, "ogl_amd", "csrc", "host_matrix.cpp"),
           os.path.join(ROOT, "ogl_amd", "csrc", "common.cpp"), "-lpthread", "-ldl", "-o", str(exe)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and "sanitize" in p.stderr and "unrecognized" in p.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert p.returncode == 0, p.stderr[-3000:]
    cases = {"blocks_even": synthetic.multi_block_case([60, 40], 48, 30), "blocks_odd": synthetic.multi_block_case([30, 17], 24, 20),
             "band_odd": synthetic.random_global_case(1029, 3, 4, seed=8), "box": synthetic.poisson_case(20),
             "one": synthetic.poisson_block(1, 1, 1), "octree": synthetic.octree_case(16, 1.5), "voronoi": synthetic.voronoi_case(3000)}
    files = []
    for name, case in cases.items():
        d, loc, _, _ = capi.host_pattern(case)
        rp = np.concatenate([[0], np.cumsum(np.bincount(loc[0], minlength=d.n_rows))]).astype(np.int32)
        cols = np.ascontiguousarray(loc[1], np.int32)
        f = tmp_path / f"{name}.bin"
        with open(f, "wb") as fh:
            np.array([d.n_rows, cols.size], np.int32).tofile(fh)
            rp.tofile(fh)
            cols.tofile(fh)
        files.append(str(f))
    r = subprocess.run([str(exe), *files], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("symx rc=0") == len(cases) and "ERROR" not in r.stderr
    # ... and the numbering code (reverse Cuthill-McKee, Hilbert order, the renumbered pattern with and without centres)
    assert r.stdout.count("numbering rcm rc=0") == len(cases) and r.stdout.count("hilbert rc=0") == len(cases), r.stdout[-2000:]
