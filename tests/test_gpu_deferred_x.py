"""GKOCG defers `x += t p` into the next turn's step_1x kernel (kernels_krylov.hip) and flushes it after a
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


# The three shapes of a GKOCG turn (kernels_krylov.hip, kernels_spmv_sym.hip): small systems fold the finalisers into the step kernels (the check of
# a turn runs at the head of the next one) -- 3 launches, or 2 on half storage, where step_1x and the SpMV are one
# kernel (k_cg_turn_sym: p_new recomputed at the gathered columns); larger systems run the five-launch turn, or with
# property fusedTurnBig the same merge between the single-workgroup finalisers (four launches, k_cg_turn_sym_big)
TURNS = {"two_launch": (1.0, 1.0, 0.0), "three_launch": (1.0, 0.0, 0.0), "four_launch": (0.0, 0.0, 1.0),
         "five_launch": (0.0, 0.0, 0.0)}


def set_turn(s, turn):
    fused, merged, merged_big = TURNS[turn]
    s.set_property("fusedFinalizers", fused)
    s.set_property("fusedTurn", merged)
    s.set_property("fusedTurnBig", merged_big)
    return s


def check_turn(s, turn, small=True):
    fused, merged, merged_big = TURNS[turn]
    assert s.get_property("fusedFinalizersInUse") == (fused if small else 0.0)
    half = s.get_property("symmetricHalf") == 1.0 and s.get_property("symmetricHalfPerChunk") == 0.0
    want = (merged if small and fused else merged_big) if half else 0.0
    assert s.get_property("fusedTurnInUse") == want


@pytest.mark.parametrize("turn", list(TURNS))
@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
@pytest.mark.parametrize("max_iter", [1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 34, 50])
def test_stop_by_max_iter_anywhere_in_the_batches(reg, oracle, system, precond, max_iter, turn):
    case, b, A, inv = system
    kw = dict(tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    s = set_turn(reg.solver(f"dx_{precond}_{turn}", capi.default_config(
        solver=capi.SOLVER_CG, preconditioner=precond, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)).set_matrix(case), turn)
    x, perf = s.solve(b, np.zeros_like(b))
    check_turn(s, turn)
    if turn == "two_launch":
        assert s.get_property("fusedTurnInUse") == 1.0   # (the box is banded: half storage)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv if precond else None, **kw)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x)


@pytest.mark.parametrize("turn", list(TURNS))
@pytest.mark.parametrize("tol", [1e-1, 1e-3, 1e-6, 1e-10])
def test_stop_by_tolerance(reg, oracle, system, tol, turn):
    case, b, A, inv = system
    kw = dict(tolerance=tol, rel_tol=0.0, max_iter=500)
    s = set_turn(reg.solver(f"dx_tol_{turn}", capi.default_config(
        solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
        update_init_guess=1, **kw)).set_matrix(case), turn)
    x, perf = s.solve(b, np.zeros_like(b))
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(x, ref.x)


@pytest.mark.parametrize("turn", list(TURNS))
def test_converged_initial_guess_leaves_x_alone(reg, oracle, system, turn):
    case, b, A, inv = system
    kw = dict(tolerance=1e-6, rel_tol=0.0, max_iter=100)
    s = set_turn(reg.solver(f"dx_conv_{turn}", capi.default_config(
        solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0, **kw)).set_matrix(case), turn)
    x1, _ = s.solve(b, np.zeros_like(b))
    cfg2 = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1,
                               adapt_min_iter=0, update_init_guess=1, **kw)
    s2 = set_turn(reg.solver(f"dx_conv2_{turn}", cfg2).set_matrix(case), turn)
    x2, perf2 = s2.solve(b, x1.copy())
    assert perf2.n_iterations == 1          # the initial check already stops
    np.testing.assert_array_equal(x2, x1)


@pytest.mark.parametrize("turn", list(TURNS))
def test_hipgraph_replay_gives_the_same_bits(reg, oracle, system, turn):
    """property hipGraph: full batches of 16 turns are captured once and replayed; the stop may fall
    anywhere inside a replayed batch.  (The fused-finaliser turns run their first batch direct: the first
    check of a solve differs from the later ones; the two p buffers of the 2-launch turn are back in place after
    a batch of 16.)"""
    case, b, A, inv = system
    for max_iter in (16, 17, 40, 64, 65):
        kw = dict(tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
        s = set_turn(reg.solver(f"dx_graph_{turn}", capi.default_config(
            solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
            update_init_guess=1, **kw)).set_matrix(case), turn)
        s.set_property("hipGraph", 1.0)
        x, perf = s.solve(b, np.zeros_like(b))
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
        assert perf.n_iterations == ref.n_iterations
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(x, ref.x)


@pytest.mark.parametrize("shape", [(16, 16, 16), (33, 31, 29), (64, 64, 64), (80, 81, 80), (1, 1, 1), (700, 1, 1), (40, 30, 1), (31, 40, 1)])
def test_fused_turns_same_bits_as_the_five_launch_turn(reg, oracle, shape):
    """Every workgroup reducing the partials itself (256 threads walking the finaliser's 1024-thread tree) and the
    check moved to the head of the next kernel change nothing: history, x, counters, with evalFrequency > 1 and
    minIter in play, up to the largest system the fused turn takes (1024 chunks) and on the first one it leaves
    to the five-launch turn; the same for the 2-launch turn (every distance table the half storage has: 1 / 2 / 3
    distances, even and odd line lengths, with and without 1/d)."""
    case = synthetic.poisson_block(*shape)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    variants = [dict(tolerance=1e-9, rel_tol=0.0, max_iter=60), dict(tolerance=1e-12, rel_tol=1e-4, max_iter=200),
                dict(tolerance=1e-7, rel_tol=0.0, max_iter=90, eval_frequency=3, min_iter=7),
                dict(tolerance=0.0, rel_tol=0.0, max_iter=23, preconditioner=capi.PRECOND_NONE)]
    for i, kw in enumerate(variants):
        out = []
        for turn in TURNS:
            base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
                        update_init_guess=1)
            base.update(kw)
            s = set_turn(reg.solver(f"ff_{i}_{turn}", capi.default_config(**base)).set_matrix(case), turn)
            x, perf = s.solve(b, np.zeros_like(b))
            n_chunks = -(-case.n_cells // capi.lib().ogl_reduction_chunk_rows())
            check_turn(s, turn, small=n_chunks <= 1024)
            out.append((x, s.history().copy(), perf.n_iterations, perf.n_norm_evals, perf.initial_residual,
                        perf.final_residual))
        for o in out[:-1]:
            np.testing.assert_array_equal(o[1], out[-1][1], err_msg=str((shape, kw)))
            np.testing.assert_array_equal(o[0], out[-1][0], err_msg=str((shape, kw)))
            assert o[2:] == out[-1][2:], (shape, kw, o[2:], out[-1][2:])
