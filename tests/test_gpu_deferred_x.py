"""GKOCG defers `x += t p` into the next turn's step_1x kernel (kernels.hip) and flushes it after a
stop.  Wherever the criterion stops -- before the first turn, in the middle of an enqueued batch of 16
turns, at a batch boundary, or with the check of the last turn the host enqueues (maxIter) -- x must
hold the same bits as the oracle's."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


@pytest.fixture(scope="module")
def system(oracle):
    case = synthetic.poisson_case(14)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    return case, b, A, oracle.jacobi_generate_scalar(rp, cols, vals)


@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
@pytest.mark.parametrize("max_iter", [1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 34, 50])
def test_stop_by_max_iter_anywhere_in_the_batches(reg, oracle, system, precond, max_iter):
    case, b, A, inv = system
    kw = dict(tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    s = reg.solver(f"dx_{precond}", capi.default_config(
        solver=capi.SOLVER_CG, preconditioner=precond, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv if precond else None, **kw)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x)


@pytest.mark.parametrize("tol", [1e-1, 1e-3, 1e-6, 1e-10])
def test_stop_by_tolerance(reg, oracle, system, tol):
    case, b, A, inv = system
    kw = dict(tolerance=tol, rel_tol=0.0, max_iter=500)
    s = reg.solver("dx_tol", capi.default_config(
        solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
        update_init_guess=1, **kw)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(x, ref.x)


def test_converged_initial_guess_leaves_x_alone(reg, oracle, system):
    case, b, A, inv = system
    kw = dict(tolerance=1e-6, rel_tol=0.0, max_iter=100)
    s = reg.solver("dx_conv", capi.default_config(
        solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0, **kw)).set_matrix(case)
    x1, _ = s.solve(b, np.zeros_like(b))
    cfg2 = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1,
                               adapt_min_iter=0, update_init_guess=1, **kw)
    s2 = reg.solver("dx_conv2", cfg2).set_matrix(case)
    x2, perf2 = s2.solve(b, x1.copy())
    assert perf2.n_iterations == 1          # the initial check already stops
    np.testing.assert_array_equal(x2, x1)


def test_hipgraph_replay_gives_the_same_bits(reg, oracle, system):
    """property hipGraph: full batches of 16 turns are captured once and replayed; the stop may fall
    anywhere inside a replayed batch."""
    case, b, A, inv = system
    for max_iter in (16, 17, 40, 64, 65):
        kw = dict(tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
        s = reg.solver("dx_graph", capi.default_config(
            solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
            update_init_guess=1, **kw)).set_matrix(case)
        s.set_property("hipGraph", 1.0)
        x, perf = s.solve(b, np.zeros_like(b))
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
        assert perf.n_iterations == ref.n_iterations
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(x, ref.x)
