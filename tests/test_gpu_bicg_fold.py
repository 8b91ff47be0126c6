"""GKOBiCGStab on small single-rank systems folds its three finalisers into the step kernels (k_bicg_fold1/2/3,
kernels_krylov.hip: 5 launches per turn instead of 8).  Wherever the criterion stops -- at the head of a turn, at the mid-turn
check on s (bicgstab::finalize then applies x += alpha y), by maxIter anywhere in the batches of 8 turns, by tolerance,
before the first turn -- history, x, counters and residuals must hold the same bits as the oracle's and as the
8-launch turn's.  Reference: Solver/BiCGStab/GKOBiCGStab.H:16-117, StoppingCriterion/StoppingCriterion.C:71-151.
"""
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
    case = synthetic.poisson_case(14, symmetric=False)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    return case, b, A, (rp, cols, vals)


def precond_of(oracle, csr, pc, k=1):
    rp, cols, vals = csr
    if pc == capi.PRECOND_NONE:
        return None
    if pc == capi.PRECOND_BJ:
        return oracle.jacobi_generate_scalar(rp, cols, vals) if k == 1 else oracle.Precond(rp, cols, vals, k)
    return oracle.Precond(rp, cols, vals, isai="spd" if pc == capi.PRECOND_ISAI else "general")


def solve_both(reg, name, case, b, **kw):
    out = []
    for fold in (1.0, 0.0):
        s = reg.solver(f"{name}_{int(fold)}", capi.default_config(
            solver=capi.SOLVER_BICGSTAB, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)).set_matrix(case)
        s.set_property("bicgFold", fold)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("fusedFinalizersInUse") == fold
        out.append((x, s.history().copy(), perf))
    return out


@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE, capi.PRECOND_GISAI])
@pytest.mark.parametrize("max_iter", [1, 2, 3, 7, 8, 9, 10, 15, 16, 17, 25])
def test_stop_by_max_iter_anywhere_in_the_batches(reg, oracle, system, precond, max_iter):
    case, b, A, csr = system
    kw = dict(tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    (x, hist, perf), (x8, hist8, perf8) = solve_both(reg, f"bf_{precond}", case, b, preconditioner=precond, **kw)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), precond_of(oracle, csr, precond), **kw)
    assert perf.n_iterations == ref.n_iterations // 2 == perf8.n_iterations
    np.testing.assert_array_equal(hist, ref.history)
    np.testing.assert_array_equal(x, ref.x)
    np.testing.assert_array_equal(hist8, hist)
    np.testing.assert_array_equal(x8, x)


@pytest.mark.parametrize("tol", [3e-1, 1e-1, 3e-2, 1e-2, 1e-3, 1e-5, 1e-8, 1e-11])
def test_stop_by_tolerance_at_either_check(reg, oracle, system, tol):
    """A tolerance sweep stops some solves at the head-of-turn check and some at the mid-turn check on s, where
    bicgstab::finalize applies x += alpha y inside the folded kernel."""
    case, b, A, csr = system
    kw = dict(tolerance=tol, rel_tol=0.0, max_iter=400)
    (x, hist, perf), (x8, hist8, perf8) = solve_both(reg, "bf_tol", case, b, preconditioner=capi.PRECOND_BJ, **kw)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), precond_of(oracle, csr, capi.PRECOND_BJ), **kw)
    assert perf.n_iterations == ref.n_iterations // 2
    assert perf.final_residual == ref.history[-1] == perf8.final_residual
    np.testing.assert_array_equal(hist, ref.history)
    np.testing.assert_array_equal(x, ref.x)
    np.testing.assert_array_equal(x8, x)
    print(f"tolerance {tol:g}: stopped at check {ref.history.size - 1} ({'mid-turn' if ref.history.size % 2 == 0 else 'head of turn'})")


def test_both_stop_positions_are_covered(oracle, system):
    case, b, A, csr = system
    inv = precond_of(oracle, csr, capi.PRECOND_BJ)
    parities = set()
    for tol in (3e-1, 1e-1, 3e-2, 1e-2, 1e-3, 1e-5, 1e-8, 1e-11):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), inv, tolerance=tol, rel_tol=0.0, max_iter=400)
        parities.add(ref.history.size % 2)
    assert parities == {0, 1}


def test_converged_initial_guess_leaves_x_alone(reg, system):
    case, b, A, csr = system
    kw = dict(tolerance=1e-6, rel_tol=0.0, max_iter=200, preconditioner=capi.PRECOND_BJ)
    (x1, _, _), _ = solve_both(reg, "bf_conv", case, b, **kw)
    s2 = reg.solver("bf_conv2", capi.default_config(solver=capi.SOLVER_BICGSTAB, export_res=1, adapt_min_iter=0,
                                                     update_init_guess=1, **kw)).set_matrix(case)
    x2, perf2 = s2.solve(b, x1.copy())
    assert s2.get_property("fusedFinalizersInUse") == 1.0
    assert perf2.n_iterations == 0 and s2.history().size == 1      # the initial check already stops (1 check / 2)
    np.testing.assert_array_equal(x2, x1)


@pytest.mark.parametrize("shape", [(16, 16, 16), (33, 31, 29), (64, 64, 64), (80, 81, 80), (1, 1, 1), (700, 1, 1), (40, 30, 1)])
def test_folded_turn_same_bits_as_the_eight_launch_turn(reg, oracle, shape):
    """Up to the largest system the folded turn takes (1024 chunks = 64^3) and on the first one it leaves alone
    (80 x 81 x 80), with evalFrequency > 1 and minIter in play, with block Jacobi and ISAI."""
    case = synthetic.poisson_block(*shape, symmetric=False, off_upper=-0.9, off_lower=-1.1)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    n_chunks = -(-case.n_cells // capi.lib().ogl_reduction_chunk_rows())
    variants = [dict(tolerance=1e-9, rel_tol=0.0, max_iter=60, preconditioner=capi.PRECOND_BJ),
                dict(tolerance=1e-12, rel_tol=1e-4, max_iter=200, preconditioner=capi.PRECOND_BJ),
                dict(tolerance=1e-7, rel_tol=0.0, max_iter=90, eval_frequency=3, min_iter=7, preconditioner=capi.PRECOND_BJ),
                dict(tolerance=0.0, rel_tol=0.0, max_iter=23, preconditioner=capi.PRECOND_NONE),
                dict(tolerance=1e-9, rel_tol=0.0, max_iter=40, preconditioner=capi.PRECOND_BJ, max_block_size=4),
                dict(tolerance=1e-9, rel_tol=0.0, max_iter=40, preconditioner=capi.PRECOND_ISAI)]
    for i, kw in enumerate(variants):
        out = []
        for fold in (1.0, 0.0):
            s = reg.solver(f"bff_{i}_{int(fold)}", capi.default_config(
                solver=capi.SOLVER_BICGSTAB, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)).set_matrix(case)
            s.set_property("bicgFold", fold)
            x, perf = s.solve(b, np.zeros_like(b))
            assert s.get_property("fusedFinalizersInUse") == (fold if n_chunks <= 1024 else 0.0)
            out.append((x, s.history().copy(), perf.n_iterations, perf.n_norm_evals, perf.initial_residual,
                        perf.final_residual))
        np.testing.assert_array_equal(out[0][1], out[1][1], err_msg=str((shape, kw)))
        np.testing.assert_array_equal(out[0][0], out[1][0], err_msg=str((shape, kw)))
        assert out[0][2:] == out[1][2:], (shape, kw, out[0][2:], out[1][2:])


# ---------------------------------------------------------------------------------------------------------------
# GKOGMRES: the finaliser between two Gram-Schmidt links folded into the next link's kernel (k_gmres_mgs_fold):
# one launch per link instead of two.  Solver/GMRES/GKOGMRES.H:13-113.
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape,sym", [((16, 16, 16), True), ((33, 31, 29), False), ((64, 64, 64), True),
                                       ((80, 81, 80), True), ((1, 1, 1), True), ((700, 1, 1), False)])
def test_gmres_folded_links_same_bits(reg, oracle, shape, sym):
    kw_case = {} if sym else dict(symmetric=False, off_upper=-0.9, off_lower=-1.1)
    case = synthetic.poisson_block(*shape, **kw_case)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    n_chunks = -(-case.n_cells // capi.lib().ogl_reduction_chunk_rows())
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    variants = [dict(tolerance=1e-9, rel_tol=0.0, max_iter=45, krylov_dim=10, preconditioner=capi.PRECOND_BJ),
                dict(tolerance=0.0, rel_tol=0.0, max_iter=23, krylov_dim=30, preconditioner=capi.PRECOND_NONE),
                dict(tolerance=1e-8, rel_tol=0.0, max_iter=40, krylov_dim=7, preconditioner=capi.PRECOND_BJ, max_block_size=4)]
    for i, kw in enumerate(variants):
        out = []
        for fold in (1.0, 0.0):
            s = reg.solver(f"gf_{i}_{int(fold)}", capi.default_config(
                solver=capi.SOLVER_GMRES, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)).set_matrix(case)
            s.set_property("gmresFold", fold)
            x, perf = s.solve(b, np.zeros_like(b))
            assert s.get_property("fusedFinalizersInUse") == (fold if n_chunks <= 1024 else 0.0)
            out.append((x, s.history().copy(), perf.n_iterations, perf.n_norm_evals, perf.initial_residual,
                        perf.final_residual))
        np.testing.assert_array_equal(out[0][1], out[1][1], err_msg=str((shape, kw)))
        np.testing.assert_array_equal(out[0][0], out[1][0], err_msg=str((shape, kw)))
        assert out[0][2:] == out[1][2:], (shape, kw)
        if i < 2 and case.n_cells <= 40000:          # ... and the oracle's bits
            P = oracle.Precond(rp, cols, vals, 1) if kw["preconditioner"] == capi.PRECOND_BJ else None
            okw = {k: v for k, v in kw.items() if k in ("tolerance", "rel_tol", "max_iter", "krylov_dim")}
            with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
                ref = oracle.gmres(A, b, np.zeros_like(b), P, **okw)
            np.testing.assert_array_equal(out[0][1], ref.history)
            np.testing.assert_array_equal(out[0][0], ref.x)


@pytest.mark.parametrize("precond,block", [(capi.PRECOND_BJ, 1), (capi.PRECOND_NONE, 1), (capi.PRECOND_BJ, 4)])
@pytest.mark.parametrize("krylov_dim", [1, 3, 5])
def test_gmres_stops_anywhere_around_the_restarts(reg, oracle, system, precond, block, krylov_dim):
    """The turn's criterion check runs at the end of the finaliser before it (the restart's before the first turn, else the
    last column's: FinArgs::check_after), and with scalar Jacobi a new basis vector is divided by its norm only at the head
    of the next turn, in the pass that applies the preconditioner to it (k_gmres_scale_mul) -- V_0 after a restart
    included.  maxIter at, one before and one after every cycle boundary, the first turn, Krylov dimension 1 (a restart
    every turn), and stops by tolerance inside a cycle: history, x and counters hold the oracle's bits, folded Gram-Schmidt
    links or not.  Solver/GMRES/GKOGMRES.H:13-113, StoppingCriterion/StoppingCriterion.C:71-151."""
    case, b, A, (rp, cols, vals) = system
    P = None if precond == capi.PRECOND_NONE else oracle.Precond(rp, cols, vals, block)
    runs = [dict(tolerance=0.0, rel_tol=0.0, max_iter=mi) for mi in range(1, 3 * krylov_dim + 3)]
    runs += [dict(tolerance=tol, rel_tol=0.0, max_iter=60) for tol in (3e-1, 3e-2, 1e-3)]
    for i, kw in enumerate(runs):
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.gmres(A, b, np.zeros_like(b), P, krylov_dim=krylov_dim, **kw)
        for fold in (1.0, 0.0):
            s = reg.solver(f"gr_{precond}_{block}_{krylov_dim}_{int(fold)}", capi.default_config(
                solver=capi.SOLVER_GMRES, preconditioner=precond, max_block_size=block, krylov_dim=krylov_dim, export_res=1,
                adapt_min_iter=0, update_init_guess=1, **kw)).set_matrix(case)
            s.set_property("gmresFold", fold)
            x, perf = s.solve(b, np.zeros_like(b))
            assert perf.n_iterations == ref.n_iterations, (kw, fold)
            np.testing.assert_array_equal(s.history(), ref.history, err_msg=str((kw, fold)))
            np.testing.assert_array_equal(x, ref.x, err_msg=str((kw, fold)))
