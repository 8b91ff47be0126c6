"""GPU parity tests (run with `-m gpu` on an MI355X): every call goes through the C ABI of
libogl_amd.so and is compared with the CPU oracle on the same seeded inputs.

Bars (task statement ③):
  * integer / index work (pattern, ldu_mapping)            : bit-exact
  * coefficient permutation, SpMV                            : bit-exact (same row order as the
    reference executor; -ffp-contract=off on both sides)
  * reductions, CG history vs the oracle run in the device's reduction tree : bit-exact
  * CG history vs the oracle in the reference executor's SEQUENTIAL order   : 1e-12 relative
    on the first checks, 1e-11 while the residual is above 1e-3 of its start, then iteration
    count +-1 and the solution to 1e-9: the two summation orders differ at rounding level and
    CG amplifies that as it converges (the same happens between two Ginkgo executors).
"""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix, rel_dev

pytestmark = pytest.mark.gpu

SEED = 20241016


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


@pytest.fixture(scope="module")
def chunk_rows():
    return capi.lib().ogl_reduction_chunk_rows()


def cg_cfg(**kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_NONE, tolerance=0.0,
                rel_tol=0.0, max_iter=50, export_res=1, matrix_format=capi.FORMAT_CSR,
                adapt_min_iter=0)
    base.update(kw)
    return capi.default_config(**base)


CASES = [
    ("box5x4x3", dict(gx=5, gy=4, gz=3), True),
    ("box5x4x3_asym", dict(gx=5, gy=4, gz=3, off_upper=-0.9, off_lower=-1.1), False),
    ("periodic", dict(gx=6, gy=5, gz=4, periodic_x=True), True),
    ("periodic_asym", dict(gx=6, gy=5, gz=4, periodic_x=True, off_upper=-0.9, off_lower=-1.1), False),
    ("cube16", dict(gx=16, gy=16, gz=16), True),
    ("cube33_ragged", dict(gx=33, gy=31, gz=29), True),      # rows not a multiple of the chunk
    ("single_cell", dict(gx=1, gy=1, gz=1), True),
    ("line", dict(gx=700, gy=1, gz=1), True),
]


@pytest.mark.parametrize("name,kw,sym", CASES, ids=[c[0] for c in CASES])
def test_device_matrix_is_bit_exact(reg, oracle, name, kw, sym):
    case = synthetic.poisson_block(symmetric=sym, **kw)
    s = reg.solver("m_" + name, cg_cfg()).set_matrix(case)
    rp, cols, mp, vals = s.local_matrix()
    o_rp, o_cols, o_vals = oracle_csr(oracle, case)
    np.testing.assert_array_equal(rp, o_rp)
    np.testing.assert_array_equal(cols, o_cols)
    np.testing.assert_array_equal(vals, o_vals)


@pytest.mark.parametrize("name,kw,sym", CASES, ids=[c[0] for c in CASES])
def test_spmv_is_bit_exact(reg, oracle, name, kw, sym):
    case = synthetic.poisson_block(symmetric=sym, **kw)
    s = reg.solver("m_" + name, cg_cfg()).set_matrix(case)
    rng = np.random.default_rng(SEED)
    x = rng.uniform(-1, 1, case.n_cells)
    o_rp, o_cols, o_vals = oracle_csr(oracle, case)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(o_rp, o_cols, o_vals, x))


def test_wide_rows_take_several_lds_passes(reg, oracle):
    # one chunk of 512 rows with > SPMV_TILE non-zeros: a hub cell coupled to every other cell
    n = 6000
    lower = np.zeros(n - 1, np.int32)
    upper = np.arange(1, n, dtype=np.int32)
    rng = np.random.default_rng(SEED)
    case = synthetic.LduCase(n, lower, upper, rng.uniform(1, 2, n) + n, rng.uniform(-1, 1, n - 1),
                             rng.uniform(-1, 1, n - 1))
    s = reg.solver("hub", cg_cfg()).set_matrix(case)
    o_rp, o_cols, o_vals = oracle_csr(oracle, case)
    assert o_rp[1] - o_rp[0] == n
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(o_rp, o_cols, o_vals, x))


@pytest.mark.parametrize("n", [1, 63, 512, 513, 100003])
def test_reductions(reg, oracle, chunk_rows, n):
    case = synthetic.poisson_block(n, 1, 1)
    s = reg.solver(f"red{n}", cg_cfg()).set_matrix(case)
    rng = np.random.default_rng(SEED)
    a, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    with blocked(oracle, chunk_rows):
        assert s.reduce("dot", a, b) == oracle.dot(a, b)
        assert s.reduce("norm1", a) == oracle.norm1(a)
        assert s.reduce("sum", a) == oracle.vsum(a)
    # and against the reference executor's left-to-right sums: rounding-level agreement
    assert s.reduce("dot", a, b) == pytest.approx(oracle.dot(a, b), rel=1e-12, abs=1e-13)
    assert s.reduce("norm1", a) == pytest.approx(oracle.norm1(a), rel=1e-13)


@pytest.mark.parametrize("precond", [capi.PRECOND_NONE, capi.PRECOND_BJ], ids=["none", "BJ"])
@pytest.mark.parametrize("n", [8, 16, 32, 64])     # 64 = BASELINE.json configs[0]'s size
def test_cg_history(reg, oracle, chunk_rows, precond, n):
    case = synthetic.poisson_case(n)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    cfg = cg_cfg(preconditioner=precond, max_iter=400, tolerance=1e-12)
    s = reg.solver(f"cg{n}_{precond}", cfg).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    hist = s.history()
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    inv = oracle.jacobi_generate_scalar(rp, cols, vals) if precond else None
    kw = dict(tolerance=1e-12, rel_tol=0.0, max_iter=400)
    with blocked(oracle, chunk_rows):
        ref_b = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
    # same reduction tree: everything bit-identical
    assert perf.n_iterations == ref_b.n_iterations
    np.testing.assert_array_equal(hist, ref_b.history)
    np.testing.assert_array_equal(x, ref_b.x)
    assert perf.initial_residual == ref_b.initial_residual
    assert perf.final_residual == ref_b.final_residual
    assert perf.norm_factor == ref_b.norm_factor
    # reference executor order (sequential sums).  The two summation orders differ at rounding
    # level (~sqrt(N) eps per dot) and CG amplifies that turn by turn, so the bar is stated on
    # the part of the history where the comparison is meaningful:
    #   residual > 1e-3 of its start : 1e-11 relative (measured maxima: <= 2e-12)
    #   whole run                    : same iteration count +-1, same solution to 1e-9
    ref_s = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
    m = min(hist.size, ref_s.history.size)
    dev = rel_dev(hist[:m], ref_s.history[:m])
    early = ref_s.history[:m] > 1e-3 * ref_s.history[0]
    assert early.sum() >= 5
    assert dev[early].max() <= 1e-11
    assert dev[:5].max() <= 1e-12
    assert abs(perf.n_iterations - ref_s.n_iterations) <= 1
    np.testing.assert_allclose(x, ref_s.x, atol=1e-9, rtol=0)
    np.testing.assert_allclose(x, xs, atol=1e-7, rtol=0)


@pytest.mark.parametrize("kw,expect_iters,expect_evals", [
    (dict(max_iter=10), 11, 11),
    (dict(max_iter=10, eval_frequency=4), 13, 4),
    (dict(max_iter=100, tolerance=2.0), 1, 1),
    (dict(max_iter=100, tolerance=1.0, min_iter=5), 6, 2),
    (dict(max_iter=0), 1, 1),
    (dict(max_iter=3, min_iter=9), 10, 2),          # minIter above maxIter: runs on to minIter
    (dict(max_iter=3, min_iter=9, eval_frequency=4), 13, 2),
])
def test_criterion_bookkeeping(reg, oracle, kw, expect_iters, expect_evals):
    case = synthetic.poisson_case(8)
    b = np.ones(case.n_cells)
    s = reg.solver("crit", cg_cfg(**kw)).set_matrix(case)
    s.upload_solution(None)
    x, perf = s.solve(b, np.zeros_like(b))
    assert (perf.n_iterations, perf.n_norm_evals) == (expect_iters, expect_evals)
    A, _ = oracle_matrix(oracle, case)
    o = dict(tolerance=kw.get("tolerance", 0.0), rel_tol=0.0, max_iter=kw["max_iter"],
             min_iter=kw.get("min_iter", 0), frequency=kw.get("eval_frequency", 1))
    ref = oracle.cg(A, b, np.zeros_like(b), None, **o)
    assert (ref.n_iterations, ref.n_evals) == (expect_iters, expect_evals)


def test_rel_tol_stop(reg, oracle, chunk_rows):
    case = synthetic.poisson_case(12)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    s = reg.solver("reltol", cg_cfg(rel_tol=1e-3, max_iter=500)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    A, _ = oracle_matrix(oracle, case)
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, b, np.zeros_like(b), None, tolerance=0.0, rel_tol=1e-3, max_iter=500)
    assert perf.n_iterations == ref.n_iterations
    assert perf.final_residual == ref.final_residual < 1e-3 * perf.initial_residual


def test_persistence_across_solves(reg, oracle, chunk_rows):
    """Second construction finds the device state by field name; x is NOT re-uploaded
    (updateInitGuess false, lduLduBase.H:235) but b and the coefficients are."""
    case = synthetic.poisson_case(10)
    rng = np.random.default_rng(SEED)
    b1, b2 = rng.uniform(-1, 1, case.n_cells), rng.uniform(-1, 1, case.n_cells)
    cfg = cg_cfg(max_iter=15)
    s = reg.solver("persist", cfg).set_matrix(case)
    x1, _ = s.solve(b1, np.zeros_like(b1))
    case2 = synthetic.poisson_case(10)
    case2.diag = case.diag * 1.5
    s2 = reg.solver("persist", cfg).set_matrix(case2)           # same field -> same handle
    assert s2._h.value == s._h.value
    x2, _ = s2.solve(b2, np.full_like(b2, 123.0))                # this psi must be ignored
    with blocked(oracle, chunk_rows):
        A1, _ = oracle_matrix(oracle, case)
        r1 = oracle.cg(A1, b1, np.zeros_like(b1), None, tolerance=0.0, rel_tol=0.0, max_iter=15)
        A2, _ = oracle_matrix(oracle, case2)
        r2 = oracle.cg(A2, b2, r1.x, None, tolerance=0.0, rel_tol=0.0, max_iter=15)
    np.testing.assert_array_equal(x1, r1.x)
    np.testing.assert_array_equal(x2, r2.x)
    np.testing.assert_array_equal(s2.history(), r2.history)
    # updateInitGuess true: psi is honoured
    s3 = reg.solver("persist", cg_cfg(max_iter=15, update_init_guess=1)).set_matrix(case2)
    x3, _ = s3.solve(b2, np.zeros_like(b2))
    with blocked(oracle, chunk_rows):
        r3 = oracle.cg(A2, b2, np.zeros_like(b2), None, tolerance=0.0, rel_tol=0.0, max_iter=15)
    np.testing.assert_array_equal(x3, r3.x)


def test_update_sys_matrix_false_keeps_old_values(reg, oracle):
    case = synthetic.poisson_case(6)
    s = reg.solver("nosys", cg_cfg(update_sys_matrix=0)).set_matrix(case)
    v0 = s.local_matrix()[3]
    case2 = synthetic.poisson_case(6)
    case2.diag = case.diag * 2
    reg.solver("nosys", cg_cfg(update_sys_matrix=0)).set_matrix(case2)
    np.testing.assert_array_equal(s.local_matrix()[3], v0)
    reg.solver("nosys", cg_cfg(update_sys_matrix=1)).set_matrix(case2)
    assert not np.array_equal(s.local_matrix()[3], v0)


def test_scaling_quirks(reg, oracle, chunk_rows):
    """Default device path: `scaling` multiplies the RHS only (lduLduBase.H:244-252), the matrix is
    untouched (HostMatrix.C:634-704).  reorderOnHost: the non-symmetric host update scales the
    matrix too, the symmetric one ignores the factor (HostMatrixFreeFunctions.C:27-28)."""
    case = synthetic.poisson_case(8, symmetric=False)
    b = np.ones(case.n_cells)
    s = reg.solver("scal", cg_cfg(scaling=3.0, max_iter=5)).set_matrix(case)
    np.testing.assert_array_equal(s.local_matrix()[3], oracle_csr(oracle, case)[2])
    s = reg.solver("scal_h", cg_cfg(scaling=3.0, reorder_on_host=1, max_iter=5)).set_matrix(case)
    np.testing.assert_array_equal(s.local_matrix()[3],
                                  oracle_csr(oracle, case, host_path=True, scaling=3.0)[2])
    np.testing.assert_array_equal(s.local_matrix()[3], 3.0 * oracle_csr(oracle, case)[2])
    cs = synthetic.poisson_case(8)
    s = reg.solver("scal_hs", cg_cfg(scaling=3.0, reorder_on_host=1, max_iter=5)).set_matrix(cs)
    np.testing.assert_array_equal(s.local_matrix()[3], oracle_csr(oracle, cs)[2])
    # RHS scaling: solving with scaling=3 equals solving 3*b
    s = reg.solver("scal_rhs", cg_cfg(scaling=3.0, max_iter=20)).set_matrix(cs)
    x, _ = s.solve(b, np.zeros_like(b))
    A, _ = oracle_matrix(oracle, cs)
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, 3.0 * b, np.zeros_like(b), None, tolerance=0.0, rel_tol=0.0, max_iter=20)
    np.testing.assert_array_equal(x, ref.x)


def test_preconditioner_caching_rules(reg, oracle):
    """Preconditioner.H:384-418: stored preconditioner reused while the per-field counter > 0."""
    case = synthetic.poisson_case(6)
    cfg = cg_cfg(preconditioner=capi.PRECOND_BJ, caching=2, max_iter=3)
    b = np.ones(case.n_cells)
    r = capi.Registry()
    try:
        s = r.solver("pc", cfg).set_matrix(case)
        s.solve(b, np.zeros_like(b))
        assert s.get_property("preconditionerCaching") == 2
        s.solve(b, np.zeros_like(b))
        assert s.get_property("preconditionerCaching") == 1
        s.solve(b, np.zeros_like(b))
        assert s.get_property("preconditionerCaching") == 0
        s.solve(b, np.zeros_like(b))                        # regenerate, counter reset
        assert s.get_property("preconditionerCaching") == 2
    finally:
        r.close()


def test_errors(reg):
    s = reg.solver("err", cg_cfg())
    with pytest.raises(capi.OglError) as e:
        s.solve(np.ones(3), np.ones(3))
    assert e.value.status == capi.ERR_STATE
    with pytest.raises(capi.OglError):
        reg.solver("err2", cg_cfg(ranks_per_gpu=2))
    with pytest.raises(capi.OglError):
        reg.solver("err3", cg_cfg(matrix_format=9))
    # processor interfaces without a communicator
    case = synthetic.poisson_block(4, 4, 4, pz=2, rank=0)
    with pytest.raises(capi.OglError) as e:
        reg.solver("err4", cg_cfg()).set_matrix(case)
    assert e.value.status == capi.ERR_STATE


def test_device_id_wraps_like_the_reference():
    """ExecutorHandler.H:90-91: device_id_ % num_devices -- rank 8 on the second node of a 2 x 8 run
    (or any rank / ranksPerGPU past the local device count) must land on a local device."""
    import torch
    n_dev = torch.cuda.device_count()
    r = capi.Registry(device_id=n_dev + 0)           # -> device 0
    try:
        case = synthetic.poisson_case(4)
        x, perf = r.solver("wrap", cg_cfg(max_iter=3)).set_matrix(case).solve(np.ones(64), np.zeros(64))
        assert perf.n_iterations == 4
    finally:
        r.close()
    r = capi.Registry(device_id=5 * n_dev + (n_dev - 1))
    r.close()


def test_empty_system(reg):
    case = synthetic.LduCase(0, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0),
                             np.zeros(0), None)
    s = reg.solver("empty", cg_cfg(max_iter=3)).set_matrix(case)
    x, perf = s.solve(np.zeros(0), np.zeros(0))
    assert x.size == 0 and perf.n_iterations >= 1


# ---------------------------------------------------------------------------- GKOBiCGStab

@pytest.mark.parametrize("precond", [capi.PRECOND_NONE, capi.PRECOND_BJ], ids=["none", "BJ"])
@pytest.mark.parametrize("sym", [False, True], ids=["asym", "sym"])
@pytest.mark.parametrize("tol", [1e-3, 1e-6, 1e-9, 1e-12])
def test_bicgstab_history(reg, oracle, chunk_rows, precond, sym, tol):
    """Both criterion checks of a turn (on r, then on s) are exercised by sweeping the tolerance:
    the stop lands on either; a mid-turn stop must apply x += alpha y (bicgstab::finalize)."""
    case = synthetic.poisson_case(12, symmetric=sym)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    cfg = cg_cfg(solver=capi.SOLVER_BICGSTAB, preconditioner=precond, max_iter=300, tolerance=tol)
    s = reg.solver(f"bicg_{precond}_{sym}", cfg).set_matrix(case)
    s.upload_solution(None)
    x, perf = s.solve(b, np.zeros_like(b))
    hist = s.history()
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    inv = oracle.jacobi_generate_scalar(rp, cols, vals) if precond else None
    with blocked(oracle, chunk_rows):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), inv, tolerance=tol, rel_tol=0.0, max_iter=300)
    assert perf.n_iterations == ref.n_iterations // 2        # GKOBiCGStab.H:114
    np.testing.assert_array_equal(hist, ref.history)
    np.testing.assert_array_equal(x, ref.x)
    assert perf.final_residual == ref.final_residual < tol
    ref_s = oracle.bicgstab(A, b, np.zeros_like(b), inv, tolerance=tol, rel_tol=0.0, max_iter=300)
    # sequential-order oracle: both satisfy the same residual bar, so they agree to ~cond * tol
    np.testing.assert_allclose(x, ref_s.x, atol=max(1e-8, 1e3 * tol), rtol=0)


def test_bicgstab_stops_on_both_checks(oracle):
    # make sure the sweep above really covers an odd (check on r) and an even (check on s) stop
    case = synthetic.poisson_case(12, symmetric=False)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    A, _ = oracle_matrix(oracle, case)
    parities = set()
    for tol in (1e-3, 1e-6, 1e-9, 1e-12, 1e-2, 1e-4, 1e-5, 1e-7):
        ref = oracle.bicgstab(A, b, np.zeros_like(b), None, tolerance=tol, rel_tol=0.0, max_iter=300)
        parities.add(ref.n_iterations % 2)
    assert parities == {0, 1}


def test_bicgstab_max_iter_is_doubled(reg, oracle):
    case = synthetic.poisson_case(8, symmetric=False)
    b = np.ones(case.n_cells)
    s = reg.solver("bicg_max", cg_cfg(solver=capi.SOLVER_BICGSTAB, max_iter=7)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    A, _ = oracle_matrix(oracle, case)
    ref = oracle.bicgstab(A, b, np.zeros_like(b), None, tolerance=0.0, rel_tol=0.0, max_iter=7)
    assert ref.n_iterations == 15 and perf.n_iterations == 7      # 2*7 checks + 1, halved
    assert s.get_property("prevSolveIters_final") == 15           # raw count is what is stored


# ---------------------------------------------------------------------------- GKOGMRES

@pytest.mark.parametrize("sym", [True, False], ids=["sym", "asym"])
@pytest.mark.parametrize("precond,k", [(capi.PRECOND_NONE, 1), (capi.PRECOND_BJ, 1), (capi.PRECOND_BJ, 4)],
                         ids=["none", "BJ1", "BJ4"])
@pytest.mark.parametrize("kdim", [5, 30, 0])
def test_gmres_history(reg, oracle, chunk_rows, sym, precond, k, kdim):
    """Restarted, right-preconditioned GMRES (krylovDim 0 = Ginkgo's default 100).  The criterion
    sees the residual vector of the last restart, so its value only moves at restarts."""
    case = synthetic.poisson_case(10, symmetric=sym)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    cfg = cg_cfg(solver=capi.SOLVER_GMRES, preconditioner=precond, max_block_size=k, max_iter=250,
                 tolerance=1e-10, krylov_dim=kdim)
    s = reg.solver(f"gmres_{sym}_{precond}_{k}_{kdim}", cfg).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    hist = s.history()
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    P = oracle.Precond(rp, cols, vals, k) if precond else None
    with blocked(oracle, chunk_rows):
        ref = oracle.gmres(A, b, np.zeros_like(b), P, tolerance=1e-10, rel_tol=0.0, max_iter=250,
                           krylov_dim=kdim)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(hist, ref.history)
    np.testing.assert_array_equal(x, ref.x)
    m = kdim or 100
    # stale residual inside a cycle: the recorded value is constant between restarts
    for c in range(0, hist.size - 1, m):
        seg = hist[c + 1:c + m + 1]
        assert np.all(seg == seg[0])
    if perf.final_residual < 1e-10:
        np.testing.assert_allclose(x, xs, atol=1e-6, rtol=0)


def test_gmres_first_check_stops(reg, oracle):
    case = synthetic.poisson_case(6)
    b = np.ones(case.n_cells)
    s = reg.solver("gmres0", cg_cfg(solver=capi.SOLVER_GMRES, tolerance=2.0, krylov_dim=7)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    assert perf.n_iterations == 1 and np.all(x == 0.0)


# ---------------------------------------------------------------------------- block Jacobi, maxBlockSize > 1

@pytest.mark.parametrize("k", [2, 4, 7, 32])
@pytest.mark.parametrize("solver", ["cg", "bicgstab"])
def test_block_jacobi(reg, oracle, chunk_rows, k, solver):
    """Preconditioner.H:91-105 with maxBlockSize k: blocks found on the host, inverted on the
    device (Gauss-Jordan, partial pivoting), applied as dense block mat-vecs; bit-exact against the
    oracle's restatement (7 does not divide the 512-row chunk: blocks straddle chunks)."""
    sym = solver == "cg"
    case = synthetic.poisson_case(11, symmetric=sym)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    kind = capi.SOLVER_CG if sym else capi.SOLVER_BICGSTAB
    cfg = cg_cfg(solver=kind, preconditioner=capi.PRECOND_BJ, max_block_size=k, max_iter=300,
                 tolerance=1e-11)
    s = reg.solver(f"bj{k}_{solver}", cfg).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    P = oracle.Precond(rp, cols, vals, k)
    assert np.all(np.diff(P.block_ptrs)[:-1] == k)          # 7-pt rows all differ: pure agglomeration
    fn = oracle.cg if sym else oracle.bicgstab
    with blocked(oracle, chunk_rows):
        ref = fn(A, b, np.zeros_like(b), P, tolerance=1e-11, rel_tol=0.0, max_iter=300)
    assert perf.n_iterations == (ref.n_iterations if sym else ref.n_iterations // 2)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x)
    np.testing.assert_allclose(x, xs, atol=1e-7, rtol=0)


def test_block_jacobi_natural_blocks(reg, oracle, chunk_rows):
    # rows with identical column pattern form natural blocks: a block-diagonal matrix of dense
    # 3x3 blocks (every row of a block has the same pattern), maxBlockSize 4 -> blocks of 3
    nb = 40
    n = 3 * nb
    lower, upper = [], []
    for b in range(nb):
        r = 3 * b
        lower += [r, r, r + 1]
        upper += [r + 1, r + 2, r + 2]
    rng = np.random.default_rng(SEED)
    F = len(lower)
    case = synthetic.LduCase(n, np.array(lower, np.int32), np.array(upper, np.int32),
                             rng.uniform(4, 5, n), rng.uniform(-1, 1, F), rng.uniform(-1, 1, F))
    cfg = cg_cfg(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_BJ, max_block_size=4,
                 max_iter=50, tolerance=1e-13)
    s = reg.solver("bj_nat", cfg).set_matrix(case)
    bvec = rng.uniform(-1, 1, n)
    x, perf = s.solve(bvec, np.zeros(n))
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    P = oracle.Precond(rp, cols, vals, 4)
    assert np.all(np.diff(P.block_ptrs) == 3)
    with blocked(oracle, chunk_rows):
        ref = oracle.bicgstab(A, bvec, np.zeros(n), P, tolerance=1e-13, rel_tol=0.0, max_iter=50)
    np.testing.assert_array_equal(x, ref.x)
    assert perf.n_iterations <= 2            # the preconditioner is the exact inverse here


def test_block_jacobi_bad_size(reg):
    case = synthetic.poisson_case(4)
    s = reg.solver("bj_bad", cg_cfg(preconditioner=capi.PRECOND_BJ, max_block_size=33)).set_matrix(case)
    with pytest.raises(capi.OglError) as e:
        s.solve(np.ones(case.n_cells), np.zeros(case.n_cells))
    assert e.value.status == capi.ERR_INVALID


# ---------------------------------------------------------------------------- MatrixMarket export

def test_mtx_export(reg, oracle, tmp_path):
    """common.C:31-58 / test/data_validation.py of the reference: files exist, coordinate format,
    row-major order, Poisson sign pattern, differ when the coefficients change."""
    import hashlib
    import scipy.io
    case = synthetic.poisson_block(6, 5, 4, periodic_x=True)
    b = np.linspace(-1, 1, case.n_cells)
    s = reg.solver("pmtx", cg_cfg(max_iter=5)).set_matrix(case)
    s.solve(b, np.zeros_like(b))
    d1 = str(tmp_path / "0.05")
    s.export_system(d1)
    A = scipy.io.mmread(f"{d1}/pmtx_A_local.mtx").tocsr()
    rp, cols, vals = oracle_csr(oracle, case)
    np.testing.assert_array_equal(A.indptr, rp)
    np.testing.assert_array_equal(A.indices, cols)
    np.testing.assert_allclose(A.data, vals, rtol=1e-14)
    lines = open(f"{d1}/pmtx_A_local.mtx").read().splitlines()
    assert lines[0] == "%%MatrixMarket matrix coordinate real general"
    ent = [ln.split(" ") for ln in lines[2:]]
    keys = [(int(r), int(c)) for r, c, _ in ent]
    assert keys == sorted(keys)                                   # row-major (data_validation.py)
    for r, c, v in ent:                                           # sign bounds, Poisson analogue
        assert (float(v) > 0) if r == c else (float(v) < 0)
    np.testing.assert_allclose(scipy.io.mmread(f"{d1}/pmtx_rhs_b_.mtx").ravel(), b, rtol=1e-14)
    hist = scipy.io.mmread(f"{d1}/pmtx_res_norms.mtx").ravel()
    np.testing.assert_array_equal(hist, s.history())
    assert open(f"{d1}/pmtx_A_non_local.mtx").read().splitlines()[1] == f"{case.n_cells} 0 0"
    case.diag = case.diag * 1.25
    s.set_matrix(case)
    d2 = str(tmp_path / "0.5")
    s.export_system(d2)
    md5 = [hashlib.md5(open(f"{d}/pmtx_A_local.mtx", "rb").read()).hexdigest() for d in (d1, d2)]
    assert md5[0] != md5[1]                                       # matrices differ between times


# ---------------------------------------------------------------------------- adaptive policy

def test_adaptive_min_iter_and_frequency_between_solves(oracle, chunk_rows):
    """StoppingCriterion.H:199-209 + common.C:104-146: the second solve of a field skips the checks
    below relaxationFactor * prevSolveIters and evaluates every `frequency` turns; both come from
    the per-field properties the first solve stored."""
    case = synthetic.poisson_case(12)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    reg = capi.Registry()
    try:
        cfg = cg_cfg(adapt_min_iter=1, export_res=0, tolerance=1e-9, max_iter=300)
        s = reg.solver("adapt", cfg).set_matrix(case)
        x1, p1 = s.solve(b, np.zeros_like(b))
        prev_iters = int(s.get_property("prevSolveIters_final"))
        assert prev_iters == p1.n_iterations
        s.set_property("_prev_solve", 9.0)            # pin the (timing-derived) relative cost
        mi, fr = capi.host_adapt_criterion(cfg, prev_iters, 9.0)
        assert mi == int(prev_iters * 0.6) and fr >= 1
        s.upload_solution(None)
        x2, p2 = s.solve(b, np.zeros_like(b))
        A, _ = oracle_matrix(oracle, case)
        with blocked(oracle, chunk_rows):
            ref = oracle.cg(A, b, np.zeros_like(b), None, tolerance=1e-9, rel_tol=0.0, max_iter=300,
                            min_iter=mi, frequency=fr, export_res=False)
        assert (p2.n_iterations, p2.n_norm_evals) == (ref.n_iterations, ref.n_evals)
        assert p2.n_norm_evals < p1.n_norm_evals
        np.testing.assert_array_equal(x2, ref.x)
        # `export true` switches the adaptation off (StoppingCriterion.H:201)
        s3 = reg.solver("adapt", cg_cfg(adapt_min_iter=1, export_res=1, tolerance=1e-9, max_iter=300))
        s3.set_matrix(case)
        s3.upload_solution(None)
        x3, p3 = s3.solve(b, np.zeros_like(b))
        assert p3.n_norm_evals == p3.n_iterations
    finally:
        reg.close()


# ---------------------------------------------------------------------------- ISAI / GISAI

@pytest.mark.parametrize("precond,kind,sym,solver", [
    (capi.PRECOND_ISAI, "spd", True, "cg"),
    (capi.PRECOND_GISAI, "general", True, "cg"),
    (capi.PRECOND_GISAI, "general", False, "bicgstab"),
    (capi.PRECOND_GISAI, "general", False, "gmres"),
    (capi.PRECOND_ISAI, "spd", True, "bicgstab"),
], ids=["ISAI-cg", "GISAI-cg", "GISAI-bicgstab", "GISAI-gmres", "ISAI-bicgstab"])
def test_isai(reg, oracle, chunk_rows, precond, kind, sym, solver):
    """Preconditioner.H:225-258 (sparsityPower 1): W generated row by row on the device (dense
    solves over the row's own pattern), applied as one (GISAI) or two (ISAI: W^T W) SpMVs."""
    case = synthetic.poisson_block(11, 9, 8, symmetric=sym, periodic_x=True, off_upper=-0.9,
                                   off_lower=-0.9 if sym else -1.1)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    sk = {"cg": capi.SOLVER_CG, "bicgstab": capi.SOLVER_BICGSTAB, "gmres": capi.SOLVER_GMRES}[solver]
    cfg = cg_cfg(solver=sk, preconditioner=precond, max_iter=300, tolerance=1e-11, krylov_dim=25)
    s = reg.solver(f"isai_{kind}_{sym}_{solver}", cfg).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    P = oracle.Precond(rp, cols, vals, isai=kind)
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=300)
    with blocked(oracle, chunk_rows):
        if solver == "cg":
            ref = oracle.cg(A, b, np.zeros_like(b), P, **kw)
        elif solver == "bicgstab":
            ref = oracle.bicgstab(A, b, np.zeros_like(b), P, **kw)
        else:
            ref = oracle.gmres(A, b, np.zeros_like(b), P, krylov_dim=25, **kw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x)
    if perf.final_residual < 1e-11:
        np.testing.assert_allclose(x, xs, atol=1e-7, rtol=0)
    # the preconditioner pays: fewer checks than the unpreconditioned run of the same solver
    if solver != "gmres":
        s0 = reg.solver(f"isai0_{sym}_{solver}", cg_cfg(solver=sk, max_iter=300, tolerance=1e-11))
        s0.set_matrix(case)
        s0.upload_solution(None)           # the field may hold an earlier parametrisation's x
        _, perf0 = s0.solve(b, np.zeros_like(b))
        assert perf.n_iterations < perf0.n_iterations


@pytest.mark.parametrize("precond,kind,sym,solver,power,widest", [
    (capi.PRECOND_ISAI, "spd", True, "cg", 2, 10),          # tril(A)^2 on the 7-point box
    (capi.PRECOND_ISAI, "spd", True, "cg", 3, 20),
    (capi.PRECOND_GISAI, "general", False, "bicgstab", 2, 25),   # A^2: 25 entries, one thread per row
    (capi.PRECOND_GISAI, "general", False, "bicgstab", 3, 63),   # A^3: 63 entries, one wavefront per row
    (capi.PRECOND_GISAI, "general", True, "cg", 3, 63),
    (capi.PRECOND_GISAI, "general", False, "bicgstab", 4, 129),  # A^4: 129 entries, one workgroup per row
    (capi.PRECOND_GISAI, "general", False, "bicgstab", 5, 225),  # A^5: 225
    (capi.PRECOND_ISAI, "spd", True, "cg", 6, 84),               # tril(A)^6: 84
], ids=["ISAI-p2", "ISAI-p3", "GISAI-p2", "GISAI-p3-wide", "GISAI-p3-wide-cg", "GISAI-p4-huge", "GISAI-p5-huge",
        "ISAI-p6-huge"])
def test_isai_sparsity_power(reg, oracle, chunk_rows, precond, kind, sym, solver, power, widest):
    """Preconditioner.H:227 `sparsityPower`: W lives on the pattern of S^power.  Rows of up to 32 entries
    are solved by one thread each, wider ones (up to 64) by one wavefront each with the system in LDS, still
    wider ones (the reference hands rows beyond 32 to Ginkgo's iterative excess system) by one
    workgroup each (up to 2048) with the system in global scratch; all must give the oracle's bits (same dense solve, same
    order of operations per element)."""
    case = synthetic.poisson_case(9, symmetric=sym)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    sk = {"cg": capi.SOLVER_CG, "bicgstab": capi.SOLVER_BICGSTAB}[solver]
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=200)
    s = reg.solver(f"isaip_{kind}_{sym}_{power}", cg_cfg(solver=sk, preconditioner=precond,
                                                          sparsity_power=power, **kw)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    P = oracle.Precond(rp, cols, vals, isai=kind, sparsity_power=power)
    assert int(np.diff(P.w_rowptr).max()) == widest
    with blocked(oracle, chunk_rows):
        ref = (oracle.cg if solver == "cg" else oracle.bicgstab)(A, b, np.zeros_like(b), P, **kw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x)
    np.testing.assert_allclose(x, xs, atol=1e-7, rtol=0)
    # a larger pattern is a better preconditioner: fewer checks than with sparsityPower 1
    s1 = reg.solver(f"isaip1_{kind}_{sym}", cg_cfg(solver=sk, preconditioner=precond, **kw)).set_matrix(case)
    s1.upload_solution(None)
    _, perf1 = s1.solve(b, np.zeros_like(b))
    assert perf.n_iterations < perf1.n_iterations


def test_isai_huge_rows_in_scratch_batches(reg, oracle, chunk_rows):
    """Rows wider than 64: the dense systems share a scratch buffer in batches (property isaiScratchBytes) -- a
    budget that holds a handful of rows at a time must give the same W.  A hub cell coupled to 150 others makes
    ONE huge row (general: the hub's; spd: the hub is the last cell) among thread-sized ones."""
    case = synthetic.poisson_case(9)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=100)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    P = oracle.Precond(rp, cols, vals, isai="general", sparsity_power=4)
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, b, np.zeros_like(b), P, **kw)
    for budget in (3 * 129 * 129 * 8, 1 << 31):
        s = reg.solver(f"isai_batch_{budget}", cg_cfg(preconditioner=capi.PRECOND_GISAI, sparsity_power=4, **kw))
        s.set_property("isaiScratchBytes", float(budget))
        s.set_matrix(case)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("isaiHugeRows") > 100
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(x, ref.x)
    n = 400
    rng = np.random.default_rng(SEED)
    lower = np.concatenate([np.arange(n - 2, dtype=np.int32), np.arange(0, 300, 2, dtype=np.int32)])
    upper = np.concatenate([np.arange(1, n - 1, dtype=np.int32), np.full(150, n - 1, np.int32)])
    order = np.lexsort((upper, lower))
    hub = synthetic.LduCase(n, lower[order], upper[order], rng.uniform(200, 201, n), rng.uniform(-1, 0, order.size), None)
    bh = rng.uniform(-1, 1, n)
    A, (rp, cols, vals) = oracle_matrix(oracle, hub)
    for pk, kind in ((capi.PRECOND_GISAI, "general"), (capi.PRECOND_ISAI, "spd")):
        P = oracle.Precond(rp, cols, vals, isai=kind)
        assert int(np.diff(P.w_rowptr).max()) == 151
        s = reg.solver(f"isai_hub_{kind}", cg_cfg(preconditioner=pk, **kw)).set_matrix(hub)
        x, perf = s.solve(bh, np.zeros_like(bh))
        assert s.get_property("isaiHugeRows") == 1.0
        with blocked(oracle, chunk_rows):
            ref = oracle.cg(A, bh, np.zeros_like(bh), P, **kw)
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(x, ref.x)


def test_isai_huge_row_with_nan_coefficients_does_not_fault(reg):
    """a column of NaNs in the dense system of a huge row (diverged coefficients) won no comparison of the
    pivot search and the row swap indexed far outside the scratch -- a GPU memory fault that took the process down.  Now
    the pivot stays at row k, as in the oracle's search, and the NaN propagates into W and the criterion: the solve ends
    (maxIter) with a NaN residual and the process lives on."""
    hub = _hub_case(150)
    hub.diag[-1] = np.nan                      # the hub's own diagonal: every entry of its system's last column
    b = np.ones(hub.n_cells)
    for pk in (capi.PRECOND_GISAI, capi.PRECOND_ISAI):
        s = reg.solver(f"isai_nan_{pk}", cg_cfg(preconditioner=pk, tolerance=1e-11, rel_tol=0.0, max_iter=5)).set_matrix(hub)
        x, perf = s.solve(b, np.zeros_like(b))
        assert s.get_property("isaiHugeRows") == 1.0
        assert np.isnan(s.history()).any() and np.isnan(x).any()
    # ... and the next solve on a sane matrix is fine (the device is alive)
    ok = _hub_case(150)
    s = reg.solver("isai_nan_after", cg_cfg(preconditioner=capi.PRECOND_GISAI, tolerance=1e-11, rel_tol=0.0, max_iter=50)).set_matrix(ok)
    x, perf = s.solve(b, np.zeros_like(b))
    assert np.isfinite(x).all() and perf.final_residual < 1e-8


def _hub_case(m, seed=SEED):
    """A chain of 2 m + 50 cells whose LAST cell is coupled to m others: one row of m + 1 entries (GISAI: the hub's;
    ISAI: the hub is the last cell, so tril(A) keeps the row)."""
    n = 2 * m + 50
    rng = np.random.default_rng(seed)
    lower = np.concatenate([np.arange(n - 2, dtype=np.int32), np.arange(0, 2 * m, 2, dtype=np.int32)])
    upper = np.concatenate([np.arange(1, n - 1, dtype=np.int32), np.full(m, n - 1, np.int32)])
    order = np.lexsort((upper, lower))
    return synthetic.LduCase(n, lower[order], upper[order], rng.uniform(3 * m, 3 * m + 1, n),
                             rng.uniform(-1, 0, order.size), None)


def test_isai_row_of_701_entries(reg, oracle, chunk_rows):
    hub = _hub_case(700)
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=60)
    A, (rp, cols, vals) = oracle_matrix(oracle, hub)
    bh = np.random.default_rng(SEED).uniform(-1, 1, hub.n_cells)
    for pk, kind in ((capi.PRECOND_GISAI, "general"), (capi.PRECOND_ISAI, "spd")):
        P = oracle.Precond(rp, cols, vals, isai=kind)
        assert int(np.diff(P.w_rowptr).max()) == 701
        s = reg.solver(f"isai_hub701_{kind}", cg_cfg(preconditioner=pk, **kw)).set_matrix(hub)
        x, perf = s.solve(bh, np.zeros_like(bh))
        assert s.get_property("isaiHugeRows") == 1.0
        with blocked(oracle, chunk_rows):
            ref = oracle.cg(A, bh, np.zeros_like(bh), P, **kw)
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(x, ref.x)


def test_isai_rows_wider_than_2048_are_refused(reg):
    hub = _hub_case(2100)                                   # one row of 2101 entries
    s = reg.solver("isai_hub2101", cg_cfg(preconditioner=capi.PRECOND_GISAI)).set_matrix(hub)
    with pytest.raises(capi.OglError) as e:
        s.solve(np.ones(hub.n_cells), np.zeros(hub.n_cells))
    assert e.value.status == capi.ERR_UNSUPPORTED and "GISAI" in str(e.value) and "sparsityPower 1" in str(e.value)
    # ... and a 7-point stencil stays below the limit for every admissible sparsityPower (8: 729 = the whole 9^3 box)
    case = synthetic.poisson_case(9)
    s = reg.solver("isai_p8", cg_cfg(preconditioner=capi.PRECOND_GISAI, sparsity_power=8, max_iter=3)).set_matrix(case)
    s.solve(np.ones(case.n_cells), np.zeros(case.n_cells))
    assert s.get_property("isaiHugeRows") == case.n_cells
    s = reg.solver("isai_p0", cg_cfg(preconditioner=capi.PRECOND_GISAI, sparsity_power=0)).set_matrix(case)
    with pytest.raises(capi.OglError) as e:
        s.solve(np.ones(case.n_cells), np.zeros(case.n_cells))
    assert e.value.status == capi.ERR_INVALID
