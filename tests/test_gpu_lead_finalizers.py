"""Leader finalisation of the GKOCG turn on systems of more than 1,024 chunks (single rank): workgroup 0 of step_1x /
step_2r walks the finaliser's 1024-thread tree over all partials and publishes the scalars, the other workgroups poll its
mailbox (device_common.hpp, kernels_krylov.hip k_cg_step1x_fin<true> / k_cg_step2r_fin<true>) -- three launches per turn
instead of five.  Same tree, same scalar logic (StoppingCriterion.C:71-151 on the device), so history, iteration count
and x carry the bits of the five-launch turn and of the oracle in the device's reduction order, wherever the criterion
stops (the deferred x update included)."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_matrix

pytestmark = pytest.mark.gpu
N = 84   # 592,704 rows = 1,158 chunks: above FUSED_FIN_MAX_CHUNKS, two batches of the leader's walk are partly filled


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


def solver(reg, name, case, lead, merged=0.0, **kw):
    """lead 1: three launches per turn (merged 0) or two (merged 1: step_1x inside the SpMV kernel on half storage,
    k_cg_turn_sym<.., LEAD>); lead 0: the five- / four-launch turn with the finalisers as launches of their own."""
    cfg = capi.default_config(solver=capi.SOLVER_CG, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)
    s = reg.solver(name, cfg)
    s.set_property("leadFinalizers", lead)
    s.set_property("fusedTurnBig", merged)
    return s.set_matrix(case)


@pytest.mark.parametrize("merged", [0.0, 1.0])
@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
@pytest.mark.parametrize("max_iter", [1, 2, 16, 17, 33, 60])
def test_same_bits_as_the_five_launch_turn_and_the_oracle(reg, oracle, system, precond, max_iter, merged):
    case, b, A, inv = system
    kw = dict(preconditioner=precond, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    out = {}
    for lead in (1.0, 0.0):
        s = solver(reg, f"lead_{precond}_{lead}_{merged}", case, lead, merged if lead else 0.0, **kw)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("leadFinalizersInUse") == lead and s.get_property("fusedFinalizersInUse") == 0.0
        assert s.get_property("fusedTurnInUse") == (merged if lead else 0.0)
        out[lead] = (x, perf.n_iterations, s.history().copy(), perf.final_residual)
    assert out[1.0][1] == out[0.0][1] == max_iter + 1
    np.testing.assert_array_equal(out[1.0][2], out[0.0][2])
    np.testing.assert_array_equal(out[1.0][0], out[0.0][0])
    assert out[1.0][3] == out[0.0][3]
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv if precond else None, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    assert ref.n_iterations == out[1.0][1]
    np.testing.assert_array_equal(out[1.0][2], ref.history)
    np.testing.assert_array_equal(out[1.0][0], ref.x)


@pytest.mark.parametrize("merged", [0.0, 1.0])
@pytest.mark.parametrize("tol", [1e-2, 1e-5, 1e-9])
def test_stop_by_tolerance_and_frequency(reg, oracle, system, tol, merged):
    case, b, A, inv = system
    kw = dict(preconditioner=capi.PRECOND_BJ, tolerance=tol, rel_tol=0.0, max_iter=600, eval_frequency=3)
    got = {}
    for lead in (1.0, 0.0):
        s = solver(reg, f"lead_tol_{lead}_{merged}", case, lead, merged if lead else 0.0, **kw)
        x, perf = s.solve(b, np.zeros_like(b))
        got[lead] = (x, perf.n_iterations, perf.n_norm_evals, s.history().copy())
    assert got[1.0][1:3] == got[0.0][1:3]
    np.testing.assert_array_equal(got[1.0][3], got[0.0][3])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), inv, tolerance=tol, rel_tol=0.0, max_iter=600, frequency=3)
    assert ref.n_iterations == got[1.0][1]
    np.testing.assert_array_equal(got[1.0][0], ref.x)


@pytest.mark.parametrize("merged", [0.0, 1.0])
def test_a_second_solve_and_a_converged_guess(reg, oracle, system, merged):
    """The mailbox is cleared and the tags restart with every solve; a guess that already satisfies the criterion stops
    at the first check and leaves x alone."""
    case, b, A, inv = system
    s = solver(reg, f"lead_twice_{merged}", case, 1.0, merged, preconditioner=capi.PRECOND_BJ, tolerance=1e-8, rel_tol=0.0,
               max_iter=600)
    x1, p1 = s.solve(b, np.zeros_like(b))
    x2, p2 = s.solve(b, np.zeros_like(b))
    assert p1.n_iterations == p2.n_iterations and p1.final_residual == p2.final_residual
    np.testing.assert_array_equal(x1, x2)
    x3, p3 = s.solve(b, x1.copy())
    assert p3.n_iterations == 1
    np.testing.assert_array_equal(x3, x1)
    assert s.get_property("leadFinalizersInUse") == 1.0


@pytest.mark.parametrize("merged", [0.0, 1.0])
@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
def test_many_tiles_of_partials(reg, precond, merged):
    """4.5 M rows = 8,789 chunks: a leader stages nine partials per virtual thread.  The five-launch turn is the witness
    (itself bit-equal to the oracle at 216^3, tests/test_gpu_fullsize_oracle.py)."""
    case = synthetic.poisson_case(165)
    b = synthetic.rhs_for_x_star(case)[0]
    got = {}
    for lead in (1.0, 0.0):
        s = solver(reg, f"lead_big_{precond}_{lead}_{merged}", case, lead, merged if lead else 0.0, preconditioner=precond,
                   tolerance=0.0, rel_tol=0.0, max_iter=40)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("leadFinalizersInUse") == lead
        got[lead] = (x, perf.n_iterations, s.history().copy())
    assert got[1.0][1] == got[0.0][1] == 41
    np.testing.assert_array_equal(got[1.0][2], got[0.0][2])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])


# ---- GKOBiCGStab: k_bicg_fold1/2/3<true> (five launches per turn + the preconditioner's own instead of eight) ----
@pytest.fixture(scope="module")
def asym_system(oracle):
    case = synthetic.poisson_case(N, symmetric=False)
    b = synthetic.rhs_for_x_star(case)[0]
    A, csr = oracle_matrix(oracle, case)
    return case, b, A, csr


def bicg_precond(oracle, csr, pc):
    rp, cols, vals = csr
    if pc == capi.PRECOND_NONE:
        return None
    if pc == capi.PRECOND_BJ:
        return oracle.jacobi_generate_scalar(rp, cols, vals)
    return oracle.Precond(rp, cols, vals, isai="spd" if pc == capi.PRECOND_ISAI else "general")


def bicg_solver(reg, name, case, lead, **kw):
    cfg = capi.default_config(solver=capi.SOLVER_BICGSTAB, export_res=1, adapt_min_iter=0, update_init_guess=1, **kw)
    s = reg.solver(name, cfg)
    s.set_property("leadFinalizers", lead)
    return s.set_matrix(case)


@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE, capi.PRECOND_GISAI])
@pytest.mark.parametrize("max_iter", [1, 2, 8, 9, 17, 30])
def test_bicgstab_same_bits_as_the_eight_launch_turn_and_the_oracle(reg, oracle, asym_system, precond, max_iter):
    case, b, A, csr = asym_system
    kw = dict(preconditioner=precond, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    out = {}
    for lead in (1.0, 0.0):
        s = bicg_solver(reg, f"lead_bicg_{precond}_{lead}", case, lead, **kw)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("leadFinalizersInUse") == lead and s.get_property("fusedFinalizersInUse") == 0.0
        out[lead] = (x, perf.n_iterations, s.history().copy(), perf.final_residual)
    assert out[1.0][1] == out[0.0][1]
    np.testing.assert_array_equal(out[1.0][2], out[0.0][2])
    np.testing.assert_array_equal(out[1.0][0], out[0.0][0])
    assert out[1.0][3] == out[0.0][3]
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), bicg_precond(oracle, csr, precond), tolerance=0.0, rel_tol=0.0,
                              max_iter=max_iter)
    assert ref.n_iterations // 2 == out[1.0][1]      # (GKOBiCGStab.H:114: two checks per turn, the count is halved)
    np.testing.assert_array_equal(out[1.0][2], ref.history)
    np.testing.assert_array_equal(out[1.0][0], ref.x)


@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_ISAI])
@pytest.mark.parametrize("tol", [1e-1, 1e-4, 1e-8])
def test_bicgstab_stop_by_tolerance(reg, oracle, asym_system, tol, precond):
    """(stops at the head-of-turn check or at the mid-turn check on s: bicgstab::finalize applies x += alpha y)"""
    case, b, A, csr = asym_system
    kw = dict(preconditioner=precond, tolerance=tol, rel_tol=0.0, max_iter=400)
    got = {}
    for lead in (1.0, 0.0):
        s = bicg_solver(reg, f"lead_bicg_tol_{precond}_{lead}", case, lead, **kw)
        x, perf = s.solve(b, np.zeros_like(b))
        got[lead] = (x, perf.n_iterations, perf.n_norm_evals, s.history().copy())
    assert got[1.0][1:3] == got[0.0][1:3]
    np.testing.assert_array_equal(got[1.0][3], got[0.0][3])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), bicg_precond(oracle, csr, precond), tolerance=tol, rel_tol=0.0, max_iter=400)
    assert ref.n_iterations // 2 == got[1.0][1]
    np.testing.assert_array_equal(got[1.0][0], ref.x)


# ---- GKOGMRES: k_gmres_mgs_fold<true> (one launch per Gram-Schmidt link instead of two) ----
@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
@pytest.mark.parametrize("max_iter", [1, 7, 12, 25])
def test_gmres_same_bits_as_the_separate_finalisers_and_the_oracle(reg, oracle, asym_system, precond, max_iter):
    case, b, A, (rp, cols, vals) = asym_system
    got = {}
    for lead in (1.0, 0.0):
        cfg = capi.default_config(solver=capi.SOLVER_GMRES, krylov_dim=10, preconditioner=precond, tolerance=0.0, rel_tol=0.0,
                                  max_iter=max_iter, export_res=1, adapt_min_iter=0, update_init_guess=1)
        s = reg.solver(f"lead_gmres_{precond}_{lead}", cfg)
        s.set_property("leadFinalizers", lead)
        s.set_property("gmresLead", lead)        # (off by default: no faster than the finaliser launches it replaces)
        s.set_matrix(case)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("leadFinalizersInUse") == lead and s.get_property("fusedFinalizersInUse") == 0.0
        got[lead] = (x, perf.n_iterations, s.history().copy())
    assert got[1.0][1] == got[0.0][1]
    np.testing.assert_array_equal(got[1.0][2], got[0.0][2])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])
    P = oracle.Precond(rp, cols, vals, 1) if precond else None
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.gmres(A, b, np.zeros_like(b), P, krylov_dim=10, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    assert ref.n_iterations == got[1.0][1]
    np.testing.assert_array_equal(got[1.0][2], ref.history)
    np.testing.assert_array_equal(got[1.0][0], ref.x)


# ---- GKOCG with a materialised z (block Jacobi, ISAI): turn_cg_generic_led -- the same two kernels with z in r's place ----
@pytest.mark.parametrize("pc,k", [(capi.PRECOND_BJ, 4), (capi.PRECOND_ISAI, 1), (capi.PRECOND_GISAI, 1)])
@pytest.mark.parametrize("max_iter", [1, 2, 17, 40])
def test_cg_with_a_block_preconditioner(reg, oracle, system, pc, k, max_iter):
    case, b, A, _ = system
    rp, cols, vals = oracle_matrix(oracle, case)[1]
    got = {}
    for lead in (1.0, 0.0):
        s = solver(reg, f"lead_gen_{pc}_{k}_{lead}", case, lead, preconditioner=pc, max_block_size=k, tolerance=0.0, rel_tol=0.0,
                   max_iter=max_iter)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("leadFinalizersInUse") == lead
        got[lead] = (x, perf.n_iterations, s.history().copy(), perf.final_residual)
    assert got[1.0][1] == got[0.0][1] == max_iter + 1
    np.testing.assert_array_equal(got[1.0][2], got[0.0][2])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])
    assert got[1.0][3] == got[0.0][3]
    P = (oracle.Precond(rp, cols, vals, k) if pc == capi.PRECOND_BJ
         else oracle.Precond(rp, cols, vals, isai="spd" if pc == capi.PRECOND_ISAI else "general"))
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, np.zeros_like(b), P, tolerance=0.0, rel_tol=0.0, max_iter=max_iter)
    assert ref.n_iterations == got[1.0][1]
    np.testing.assert_array_equal(got[1.0][2], ref.history)
    np.testing.assert_array_equal(got[1.0][0], ref.x)


def test_cg_with_a_block_preconditioner_stops_by_tolerance(reg, oracle, system):
    case, b, A, _ = system
    got = {}
    for lead in (1.0, 0.0):
        s = solver(reg, f"lead_gen_tol_{lead}", case, lead, preconditioner=capi.PRECOND_BJ, max_block_size=4, tolerance=1e-7,
                   rel_tol=0.0, max_iter=500)
        x, perf = s.solve(b, np.zeros_like(b))
        x2, perf2 = s.solve(b, np.zeros_like(b))
        np.testing.assert_array_equal(x2, x)
        got[lead] = (x, perf.n_iterations, s.history().copy())
    assert got[1.0][1] == got[0.0][1]
    np.testing.assert_array_equal(got[1.0][2], got[0.0][2])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])


@pytest.mark.parametrize("merged", [0.0, 1.0])
def test_leader_turns_replayed_as_a_graph(reg, system, merged):
    """hipGraph on request for a system of this size: full batches of 16 leader turns are captured once and replayed (the tag
    of a launch comes from the scalar slots, the mailbox is cleared outside the graph): same bits as stream launches."""
    case, b, A, inv = system
    got = {}
    for graph in (1.0, 0.0):
        s = solver(reg, f"lead_graph_{merged}_{graph}", case, 1.0, merged, preconditioner=capi.PRECOND_BJ, tolerance=0.0,
                   rel_tol=0.0, max_iter=70)
        s.set_property("hipGraph", graph)
        x, perf = s.solve(b, np.zeros_like(b))
        x2, perf2 = s.solve(b, np.zeros_like(b))
        np.testing.assert_array_equal(x2, x)
        assert s.get_property("leadFinalizersInUse") == 1.0
        if graph:
            assert s.get_property("hipGraphCaptures") >= 1.0
        got[graph] = (x, perf.n_iterations, s.history().copy())
    assert got[1.0][1] == got[0.0][1] == 71
    np.testing.assert_array_equal(got[1.0][2], got[0.0][2])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])


def test_a_leader_that_never_publishes_fails_the_solve_loudly(reg, system):
    """leadTimeoutS 0: the polling workgroups give up at once -- the solve must end (no hang), report the defect by name,
    and leave the solver usable: the next solve with the default time-out gives the bits of the one before the failure."""
    case, b, _, _ = system
    s = solver(reg, "lead_timeout", case, 1.0, preconditioner=capi.PRECOND_BJ, tolerance=0.0, rel_tol=0.0, max_iter=20)
    x0, p0 = s.solve(b, np.zeros_like(b))
    h0 = s.history().copy()
    s.set_property("leadTimeoutS", 0.0)
    with pytest.raises(capi.OglError) as e:
        s.solve(b, np.zeros_like(b))
    assert "leader finalisation timed out" in str(e.value)
    s.set_property("leadTimeoutS", 10.0)
    x1, p1 = s.solve(b, np.zeros_like(b))
    assert p1.n_iterations == p0.n_iterations
    np.testing.assert_array_equal(s.history(), h0)
    np.testing.assert_array_equal(x1, x0)
