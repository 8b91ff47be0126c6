"""The STREAM instantiations of the SpMV kernels (matrix data streamed past the caches; k_spmv_sym's d = 1 lower
entry taken from the neighbouring lane by a shuffle) are chosen by the launcher above 288 MB of matrix + vectors
-- sizes the oracle cannot follow.  `streamAboveBytes` (property) / OGL_STREAM_ABOVE_BYTES (environment) move the
threshold, so the small bit-exact cases run on them too."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def cfg(**kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-11, rel_tol=0.0,
                max_iter=14, export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0)
    base.update(kw)
    return capi.default_config(**base)


def randomise(case, seed):
    rng = np.random.default_rng(seed)
    case.upper[:] = rng.uniform(-1.0, -0.25, case.upper.size)
    if case.lower is not None:
        case.lower[:] = rng.uniform(-1.0, -0.25, case.lower.size)
    case.diag[:] = rng.uniform(7.0, 9.0, case.n_cells)
    return case


# even and odd line lengths (FAST and plain instantiation of k_spmv_sym), rows not a multiple of 128 / 512,
# one wavefront's worth, a line longer than a wavefront's 128 rows, 1-D and 2-D boxes
BOXES = [(64, 32, 16), (66, 7, 5), (33, 31, 29), (130, 3, 3), (258, 2, 2), (2, 2, 2), (127, 1, 1), (1, 96, 1),
         (37, 5, 41), (20, 20, 20)]
LAYOUTS = {"sym": dict(compress_indices=1, symmetric_half=1), "sell": dict(compress_indices=1, symmetric_half=0),
           "csr": dict(compress_indices=0), "ell": dict(matrix_format=capi.FORMAT_ELL)}


def test_four_launch_turn_on_half_storage_stream_and_plain(reg, oracle):
    """k_cg_turn_sym_big (property fusedTurnBig: step_1x and the SpMV of larger systems in one kernel) in its STREAM
    and plain instantiations, every distance table of BOXES: history and x as the oracle's."""
    rng = np.random.default_rng(20241017)
    for i, (gx, gy, gz) in enumerate(BOXES):
        case = randomise(synthetic.poisson_block(gx=gx, gy=gy, gz=gz), 60 + i)
        if case.n_cells < 2:
            continue
        rp, cols, vals = oracle_csr(oracle, case)
        x = rng.uniform(-1, 1, case.n_cells)
        b = rng.uniform(-1, 1, case.n_cells)
        A, _ = oracle_matrix(oracle, case)
        for precond in (capi.PRECOND_BJ, capi.PRECOND_NONE):
            inv = oracle.jacobi_generate_scalar(rp, cols, vals) if precond else None
            with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
                ref = oracle.cg(A, b, x.copy(), inv, tolerance=1e-11, rel_tol=0.0, max_iter=19)
            for forced in (1, 0):
                s = reg.solver(f"big_{forced}_{precond}", cfg(max_iter=19, preconditioner=precond, update_init_guess=1,
                                                              **LAYOUTS["sym"]))
                s.set_property("streamAboveBytes", 0.0 if forced else 1e18)
                s.set_property("fusedFinalizers", 0.0)
                s.set_property("fusedTurnBig", 1.0)
                s.set_matrix(case)
                xs, perf = s.solve(b, x.copy())
                if s.get_property("symmetricHalfPerChunk") == 0.0:
                    assert s.get_property("fusedTurnInUse") == 1.0 and s.get_property("spmvStream") == float(forced)
                np.testing.assert_array_equal(s.history(), ref.history, err_msg=str((forced, precond, gx, gy, gz)))
                np.testing.assert_array_equal(xs, ref.x, err_msg=str((forced, precond, gx, gy, gz)))


@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_stream_kernels_same_bits_as_the_oracle(reg, oracle, layout):
    rng = np.random.default_rng(20241016)
    for i, (gx, gy, gz) in enumerate(BOXES):
        case = randomise(synthetic.poisson_block(gx=gx, gy=gy, gz=gz), 40 + i)
        rp, cols, vals = oracle_csr(oracle, case)
        out = {}
        x = rng.uniform(-1, 1, case.n_cells)
        b = rng.uniform(-1, 1, case.n_cells)
        for forced in (1, 0):
            s = reg.solver(f"stream_{layout}_{forced}", cfg(**LAYOUTS[layout]))
            s.set_property("streamAboveBytes", 0.0 if forced else 1e18)
            s.set_matrix(case)
            assert s.get_property("spmvStream") == float(forced), (layout, gx, gy, gz)
            if layout == "sym" and case.n_cells > 1:
                assert s.get_property("symmetricHalf") == 1.0
                ds = sorted({d for d, on in ((1, gx > 1), (gx, gy > 1), (gx * gy, gz > 1)) if on})
                fast = ds[0] == 1 and all(d % 2 == 0 for d in ds[1:])   # the pair-load instantiation
                assert s.get_property("spmvSymFast") == float(fast), (gx, gy, gz)
            np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x), err_msg=str((layout, gx, gy, gz)))
            xs, perf = s.solve(b, x.copy())
            out[forced] = (xs, s.history().copy())
        A, _ = oracle_matrix(oracle, case)
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.cg(A, b, x.copy(), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=1e-11,
                            rel_tol=0.0, max_iter=14)
        for forced in (1, 0):
            np.testing.assert_array_equal(out[forced][1], ref.history, err_msg=str((layout, forced, gx, gy, gz)))
            np.testing.assert_array_equal(out[forced][0], ref.x, err_msg=str((layout, forced, gx, gy, gz)))


def test_stream_kernels_irregular_and_asymmetric(reg, oracle):
    """Delta / column coded chunks, the spill list, BiCGStab's two-dot SpMV: STREAM forced on."""
    rng = np.random.default_rng(5)
    cases = {"shuffled": synthetic.renumber_case(synthetic.poisson_case(24), 512),
             "long rows": synthetic.long_rows_case(synthetic.poisson_case(24), 0.03, 24),
             "asym": randomise(synthetic.poisson_case(18, symmetric=False), 8)}
    for name, case in cases.items():
        rp, cols, vals = oracle_csr(oracle, case)
        for comp in (2, 0):
            s = reg.solver(f"stream_irr_{comp}", cfg(compress_indices=comp, renumber=capi.RENUMBER_OFF,
                                                     solver=capi.SOLVER_BICGSTAB if name == "asym" else capi.SOLVER_CG))
            s.set_property("streamAboveBytes", 0.0)
            s.set_matrix(case)
            assert s.get_property("spmvStream") == 1.0
            x = rng.uniform(-1, 1, case.n_cells)
            np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x), err_msg=name)
            b = rng.uniform(-1, 1, case.n_cells)
            xs, perf = s.solve(b, x.copy())
            A, _ = oracle_matrix(oracle, case)
            fn = oracle.bicgstab if name == "asym" else oracle.cg
            with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
                ref = fn(A, b, x.copy(), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=1e-11, rel_tol=0.0,
                         max_iter=14)
            np.testing.assert_array_equal(s.history(), ref.history, err_msg=name)
            np.testing.assert_array_equal(xs, ref.x, err_msg=name)


def test_existing_bit_exact_suites_with_stream_forced():
    """test_gpu_sym / test_gpu_sell / test_gpu_formats / test_gpu_renumber once more, every SpMV launch on its
    STREAM instantiation (environment switch, read when the library loads: hence a child process)."""
    env = dict(os.environ, OGL_STREAM_ABOVE_BYTES="0")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_gpu_sym.py", "test_gpu_sell.py", "test_gpu_formats.py",
                                                      "test_gpu_renumber.py")]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *files],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:] + "\n" + p.stderr[-2000:]
    assert " passed" in p.stdout and "skipped" not in p.stdout.splitlines()[-1], p.stdout[-500:]
