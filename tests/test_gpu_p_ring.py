"""x touched by every K-th head of the three-launch leader turn (property deferX = K search-direction buffers used in turn,
PRing in kernels.hpp, k_cg_step1x_fin<true, K>): the heads in between leave t_j p_j pending -- t_j in the device scalars,
p_j intact in its ring buffer -- and the head at ring position 0 adds the pending terms oldest first, then its own.  The
sums are those of K single updates, so x, history and counts carry the bits of the oracle (device reduction order) and of
x-every-turn, wherever the criterion stops: at every ring position, by tolerance, inside a replayed hipGraph batch."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_matrix

pytestmark = pytest.mark.gpu
N = 84   # 592,704 rows = 1,158 chunks: the leader finalisation is the default above 1,024


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


@pytest.fixture(scope="module")
def system(oracle):
    case = synthetic.poisson_case(N)
    b = synthetic.rhs_for_x_star(case)[0]
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    return case, b, A, oracle.jacobi_generate_scalar(rp, cols, vals)


def solver(reg, name, case, ring, **kw):
    cfg = capi.default_config(solver=capi.SOLVER_CG, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)
    s = reg.solver(name, cfg)
    s.set_property("deferX", float(ring))
    s.set_property("fusedTurnBig", 0.0)
    return s.set_matrix(case)


def test_default_is_a_ring_of_two(reg, system):
    case, b, _, _ = system
    cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, max_iter=5, tolerance=0.0, rel_tol=0.0)
    s = reg.solver("ring_default", cfg)
    s.set_property("fusedTurnBig", 0.0)   # (the three-launch turn: at this size the default is step_1x inside the SpMV kernel)
    s.set_matrix(case)
    s.solve(b, np.zeros_like(b))
    assert s.get_property("leadFinalizersInUse") == 1.0 and s.get_property("deferXInUse") == 2.0
    s.set_property("deferX2", 0.0)   # (the switch of the two-buffer version still turns the deferral off)
    s.solve(b, np.zeros_like(b))
    assert s.get_property("deferXInUse") == 0.0


@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
@pytest.mark.parametrize("max_iter", [3, 4, 5, 6, 7, 8, 9, 11, 19, 24, 50])
def test_stop_at_every_ring_position(reg, oracle, system, precond, max_iter):
    """tolerance 0: the criterion stops at check max_iter + 1, i.e. at the head of ring position max_iter % K."""
    case, b, A, inv = system
    kw = dict(preconditioner=precond, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv if precond else None, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    for ring in (0, 2, 4, 8):
        s = solver(reg, f"ring_{precond}_{ring}", case, ring, **kw)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("deferXInUse") == float(ring) and s.get_property("leadFinalizersInUse") == 1.0
        assert perf.n_iterations == ref.n_iterations == max_iter + 1, ring
        np.testing.assert_array_equal(s.history(), ref.history, err_msg=str(ring))
        np.testing.assert_array_equal(x, ref.x, err_msg=str(ring))


@pytest.mark.parametrize("tol,frequency", [(1e-2, 1), (1e-4, 3), (1e-6, 5), (1e-9, 7)])
def test_stop_by_tolerance(reg, oracle, system, tol, frequency):
    case, b, A, inv = system
    kw = dict(preconditioner=capi.PRECOND_BJ, tolerance=tol, rel_tol=0.0, max_iter=600, eval_frequency=frequency)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv, tolerance=tol, rel_tol=0.0, max_iter=600, frequency=frequency)
    for ring in (0, 2, 4, 8):
        s = solver(reg, f"ring_tol_{ring}", case, ring, **kw)
        x, perf = s.solve(b, np.zeros_like(b))
        assert perf.n_iterations == ref.n_iterations, ring
        np.testing.assert_array_equal(s.history(), ref.history, err_msg=str(ring))
        np.testing.assert_array_equal(x, ref.x, err_msg=str(ring))


@pytest.mark.parametrize("ring", [2, 4, 8])
def test_solves_in_a_row_with_the_previous_solution_as_the_guess(reg, oracle, system, ring):
    """Every solve starts its ring at position 0 with nothing pending, whatever the solve before left in the buffers and
    in t_ring; the device's x (updateInitGuess false) is the guess of the next solve."""
    case, b, A, inv = system
    x_ref = np.zeros_like(b)
    for max_iter in (7, 5, 10):
        cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
                                  update_init_guess=0, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
        s = reg.solver(f"ring_row_{ring}", cfg)   # (a construction per solve, as OpenFOAM does it: same field, same state)
        s.set_property("deferX", float(ring))
        s.set_matrix(case)
        x, perf = s.solve(b, np.zeros_like(b))
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.cg(A, b, x_ref, inv, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
        x_ref = ref.x
        assert perf.n_iterations == ref.n_iterations
        np.testing.assert_array_equal(s.history(), ref.history, err_msg=str(max_iter))
        np.testing.assert_array_equal(x, ref.x, err_msg=str(max_iter))


@pytest.mark.parametrize("precond,block", [(capi.PRECOND_BJ, 4), (capi.PRECOND_ISAI, 1)])
def test_materialised_preconditioner(reg, precond, block):
    """GKOCG with a preconditioner that is a product of its own (turn_cg_generic_led): the same ring; witness = x every
    turn (itself bit-equal to the oracle, tests/test_gpu_lead_finalizers.py)."""
    case = synthetic.poisson_case(N)
    b = synthetic.rhs_for_x_star(case)[0]
    got = {}
    for ring in (0, 2, 4, 8):
        s = solver(reg, f"ring_generic_{precond}_{ring}", case, ring, preconditioner=precond, max_block_size=block,
                   tolerance=1e-7, rel_tol=0.0, max_iter=400)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("deferXInUse") == float(ring)
        got[ring] = (x, perf.n_iterations, s.history().copy())
    for ring in (2, 4, 8):
        assert got[ring][1] == got[0][1]
        np.testing.assert_array_equal(got[ring][2], got[0][2], err_msg=str(ring))
        np.testing.assert_array_equal(got[ring][0], got[0][0], err_msg=str(ring))
