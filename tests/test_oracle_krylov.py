"""Oracle self-consistency for the part the reference cannot pin (Krylov arithmetic lives in
Ginkgo, absent): assembled operator vs LDU definition, CG/BiCGStab vs scipy at solution
level, criterion bookkeeping (StoppingCriterion.C:71-151).  "Parity unpinned" by the
reference -- these only guard the restatement against itself and against scipy.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from ogl_amd import synthetic


def assemble(oracle, case, host_path=False, scaling=1.0):
    ifs = [oracle.Iface(f.kind, f.face_cells, f.bou_coeffs, f.neighb_proc, f.neighb_patch)
           for f in case.interfaces]
    rows, cols, perm = oracle.init_local_sparsity_pattern(case.n_cells, case.upper_addr,
                                                          case.lower_addr, case.symmetric, ifs)
    vals = oracle.update_local_matrix_data(case.diag, case.upper, case.lower, ifs, perm,
                                           host_path=host_path, scaling=scaling)
    rowptr = oracle.rowptr_from_rows(case.n_cells, rows)
    return rowptr, cols, vals


@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("periodic", [False, True])
def test_assembled_operator_matches_ldu_definition(oracle, sym, periodic):
    case = synthetic.poisson_block(5, 4, 3, symmetric=sym, periodic_x=periodic,
                                   off_upper=-0.9, off_lower=-1.1 if not sym else -0.9)
    rowptr, cols, vals = assemble(oracle, case)
    # row-major, strictly increasing columns inside a row (data_validation.py's sortedness)
    for r in range(case.n_cells):
        c = cols[rowptr[r]:rowptr[r + 1]]
        assert np.all(np.diff(c) > 0)
    rng = np.random.default_rng(20241016)
    x = rng.uniform(-1, 1, case.n_cells)
    y = oracle.spmv(rowptr, cols, vals, x)
    np.testing.assert_allclose(y, synthetic.apply_case(case, x), rtol=1e-13, atol=1e-13)


def test_host_path_scaling_quirk(oracle):
    # reorderOnHost: non-symmetric path scales, symmetric path ignores scale
    case = synthetic.poisson_case(4, symmetric=False)
    _, _, v1 = assemble(oracle, case, host_path=True, scaling=1.0)
    _, _, v3 = assemble(oracle, case, host_path=True, scaling=3.0)
    np.testing.assert_array_equal(v3, 3.0 * v1)
    case = synthetic.poisson_case(4, symmetric=True)
    _, _, v1 = assemble(oracle, case, host_path=True, scaling=1.0)
    _, _, v3 = assemble(oracle, case, host_path=True, scaling=3.0)
    np.testing.assert_array_equal(v3, v1)
    _, _, vd = assemble(oracle, case, host_path=False)
    np.testing.assert_array_equal(vd, v1)


@pytest.mark.parametrize("precond", [False, True])
def test_cg_solution_vs_scipy(oracle, precond):
    case = synthetic.poisson_case(12)
    rowptr, cols, vals = assemble(oracle, case)
    A = sp.csr_matrix((vals, cols, rowptr), shape=(case.n_cells,) * 2)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = A @ xs
    D = oracle.DistMatrix(rowptr, cols, vals)
    inv = oracle.jacobi_generate_scalar(rowptr, cols, vals) if precond else None
    res = oracle.cg(D, b, np.zeros_like(b), inv, tolerance=1e-13, rel_tol=0.0, max_iter=500)
    x_ref, info = spla.cg(A, b, rtol=1e-14, atol=0.0, maxiter=2000)
    assert info == 0
    np.testing.assert_allclose(res.x, x_ref, rtol=0, atol=1e-9)
    np.testing.assert_allclose(res.x, xs, rtol=0, atol=1e-9)
    # history bookkeeping: one entry per check, first = initial residual, last = final
    assert res.history.size == res.n_iterations
    assert res.history[0] == res.initial_residual
    assert res.history[-1] == res.final_residual
    assert res.final_residual < 1e-13
    # normalised-L1 residual of the returned x, recomputed independently
    r = b - A @ res.x
    xbar = np.full_like(b, res.x.mean())
    nf0 = np.abs(A @ np.zeros_like(b) - A @ np.zeros_like(b)).sum() + np.abs(b).sum() + 1e-15
    assert res.norm_factor == pytest.approx(nf0, rel=1e-12)
    assert np.abs(r).sum() / res.norm_factor == pytest.approx(res.final_residual, rel=1e-3)
    del xbar


def test_cg_criterion_max_iter_and_frequency(oracle):
    case = synthetic.poisson_case(8)
    rowptr, cols, vals = assemble(oracle, case)
    D = oracle.DistMatrix(rowptr, cols, vals)
    b = np.ones(case.n_cells)
    res = oracle.cg(D, b, np.zeros_like(b), None, tolerance=0.0, rel_tol=0.0, max_iter=10)
    assert res.n_iterations == 11      # nIterations = CG steps + 1
    assert res.n_evals == 11
    res = oracle.cg(D, b, np.zeros_like(b), None, tolerance=0.0, rel_tol=0.0, max_iter=10,
                    frequency=4)
    assert res.n_iterations == 13      # stops at the first evaluated check with iter >= 10
    assert res.n_evals == 4            # iter 0, 4, 8, 12
    res = oracle.cg(D, b, np.zeros_like(b), None, tolerance=2.0, rel_tol=0.0, max_iter=100,
                    min_iter=5)
    assert res.n_iterations == 1       # the iter==0 check is never skipped (:77)
    res = oracle.cg(D, b, np.zeros_like(b), None, tolerance=1.0, rel_tol=0.0, max_iter=100,
                    min_iter=5)
    assert res.n_iterations == 6       # init residual == 1.0 is not < 1.0; next check at minIter
    assert res.n_evals == 2


def test_bicgstab_solution_vs_scipy(oracle):
    case = synthetic.poisson_case(10, symmetric=False)
    rowptr, cols, vals = assemble(oracle, case)
    A = sp.csr_matrix((vals, cols, rowptr), shape=(case.n_cells,) * 2)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = A @ xs
    D = oracle.DistMatrix(rowptr, cols, vals)
    inv = oracle.jacobi_generate_scalar(rowptr, cols, vals)
    res = oracle.bicgstab(D, b, np.zeros_like(b), inv, tolerance=1e-13, rel_tol=0.0, max_iter=500)
    np.testing.assert_allclose(res.x, xs, rtol=0, atol=1e-9)
    assert res.final_residual < 1e-13


def test_blocked_reduction_equals_sequential_to_rounding(oracle):
    rng = np.random.default_rng(20241016)
    a, b = rng.uniform(-1, 1, 100003), rng.uniform(-1, 1, 100003)
    s = oracle.dot(a, b)
    oracle.set_reduction(oracle.REDUCE_BLOCKED, 512)
    try:
        t = oracle.dot(a, b)
        n1 = oracle.norm1(a)
    finally:
        oracle.set_reduction(oracle.REDUCE_SEQUENTIAL)
    assert t == pytest.approx(s, rel=1e-12, abs=1e-12)
    assert n1 == pytest.approx(np.abs(a).sum(), rel=1e-13)


def test_adaptive_policy(oracle):
    # StoppingCriterion.H:199-209
    assert oracle.adapt_criterion(0, 1, True, 50, prev_rel_cost=4.0) == (0, 1)      # export: off
    assert oracle.adapt_criterion(0, 1, False, 50, prev_rel_cost=0.0) == (0, 1)     # cost 0: off
    mi, fr = oracle.adapt_criterion(0, 1, False, 100, prev_rel_cost=1.0)
    assert mi == 60 and fr == int(1 / np.sqrt(1.0 / (100 * 0.4)))
    mi, fr = oracle.adapt_criterion(0, 1, False, 100000, prev_rel_cost=1.0)
    assert fr == 100                                                                # normEvalLimit


@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("power", [1, 2, 3, 5])       # 5: rows wider than 64 (general: 216 = the whole 6^3 box)
def test_isai_defining_property_with_sparsity_power(oracle, sym, power):
    """Preconditioner.H:227 `sparsityPower`: W lives on the pattern of S^power (S = tril(A) for ISAI, A for
    GISAI).  General: (W A)(i, j) = delta_ij for every (i, j) of W's pattern -- the defining property of
    the incomplete sparse approximate inverse.  SPD: W is lower triangular and (W A W^T)(i, i) = 1."""
    import scipy.sparse as sp
    from ogl_amd import synthetic
    from helpers import oracle_csr
    case = synthetic.poisson_case(6, symmetric=sym)
    rp, cols, vals = oracle_csr(oracle, case)
    A = sp.csr_matrix((vals, cols, rp))
    P = oracle.Precond(rp, cols, vals, isai="spd" if sym else "general", sparsity_power=power)
    nnz = P.w_rowptr[-1]
    W = sp.csr_matrix((P.w_vals[:nnz], P.w_cols[:nnz], P.w_rowptr))
    S = sp.tril(A) if sym else A
    pat = (abs(S) ** power).tocsr()
    pat.sum_duplicates()
    assert np.array_equal(pat.indptr, P.w_rowptr) and np.array_equal(pat.indices, P.w_cols[:nnz])
    if sym:
        assert sp.triu(W, 1).nnz == 0
        np.testing.assert_allclose((W @ A @ W.T).diagonal(), 1.0, rtol=1e-12)
    else:
        R = (W @ A - sp.identity(A.shape[0])).toarray()
        assert np.abs(R[(W != 0).toarray()]).max() < 1e-13
    n = 4300                                                    # a hub cell coupled to 2100 others: a row of 2101 entries
    lower = np.concatenate([np.arange(n - 2), np.arange(0, 4200, 2)]).astype(np.int32)
    upper = np.concatenate([np.arange(1, n - 1), np.full(2100, n - 1)]).astype(np.int32)
    order = np.lexsort((upper, lower))
    hub = synthetic.LduCase(n, lower[order], upper[order], np.full(n, 6300.0), np.full(order.size, -1.0), None)
    rph, colsh, valsh = oracle_csr(oracle, hub)
    with pytest.raises(ValueError):
        oracle.Precond(rph, colsh, valsh, isai="general")                      # rows wider than 2048


def test_openmp_baseline_follows_the_sequential_oracle(oracle):
    """cpu_baseline_omp's solver (passes fused as the GPU kernels fuse them, parallel reductions) against the
    sequential restatement: same iteration counts, same iterates up to the order of the sums; stops by tolerance,
    by maxIter and at the initial check."""
    import numpy as np
    from ogl_amd import synthetic
    from helpers import oracle_matrix
    case = synthetic.poisson_case(18)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    for inv in (oracle.jacobi_generate_scalar(rp, cols, vals), None):
        for kw in (dict(tolerance=1e-10, rel_tol=0.0, max_iter=300), dict(tolerance=0.0, rel_tol=0.0, max_iter=17),
                   dict(tolerance=1e3, rel_tol=0.0, max_iter=17)):
            ref = oracle.cg(A, b, np.zeros_like(b), inv, **kw)
            got = oracle.cg_omp(A, b, np.zeros_like(b), inv, **kw)
            assert got.n_iterations == ref.n_iterations
            np.testing.assert_allclose(got.x, ref.x, rtol=0, atol=1e-11)
            # (the sums run in another order: the difference grows along the recurrence, as for any parallel reduction)
            big = np.asarray(ref.history) > 1e-5 * ref.history[0]
            np.testing.assert_allclose(np.asarray(got.history)[big], np.asarray(ref.history)[big], rtol=1e-8)
            np.testing.assert_allclose(got.history[:10], ref.history[:10], rtol=1e-12)


@pytest.mark.parametrize("kind", ["bj4", "isai"])
def test_preconditioner_of_the_callers_numbering_on_a_permuted_system(oracle, kind):
    """Precond(caller=...): block-Jacobi blocks / ISAI(spd)'s triangle formed on the matrix in the caller's numbering
    and carried through a permutation (what the product does under `renumber`, Preconditioner.H:91-105, :225-241):
    CG on P A P^T with it walks the same iterates as CG on A with the plain preconditioner, at rounding level."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from helpers import oracle_matrix
    from ogl_amd import synthetic
    case = synthetic.poisson_case(9)
    A0, (rp0, c0, v0) = oracle_matrix(oracle, case)
    n = case.n_cells
    new_id = np.random.default_rng(3).permutation(n).astype(np.int32)
    rp, cols, vals, _ = oracle.permute_csr(rp0, c0, v0, new_id)
    A = oracle.DistMatrix(rp, cols, vals)
    b = np.random.default_rng(4).uniform(-1, 1, n)
    bp = np.empty_like(b)
    bp[new_id] = b
    if kind == "bj4":
        P0 = oracle.Precond(rp0, c0, v0, 4)
        P = oracle.Precond(rp, cols, vals, 4, caller=(rp0, c0, v0, new_id))
        Pn = oracle.Precond(rp, cols, vals, 4)                   # blocks of the NEW numbering: another operator
    else:
        P0 = oracle.Precond(rp0, c0, v0, isai="spd")
        P = oracle.Precond(rp, cols, vals, isai="spd", caller=(rp0, c0, v0, new_id))
        Pn = oracle.Precond(rp, cols, vals, isai="spd")
    kw = dict(tolerance=1e-10, rel_tol=0.0, max_iter=300)
    r0 = oracle.cg(A0, b, np.zeros(n), P0, **kw)
    r = oracle.cg(A, bp, np.zeros(n), P, **kw)
    rn = oracle.cg(A, bp, np.zeros(n), Pn, **kw)
    assert r.n_iterations == r0.n_iterations
    np.testing.assert_allclose(r.history[:20], r0.history[:20], rtol=1e-10)
    np.testing.assert_allclose(r.x[new_id], r0.x, atol=1e-9)
    assert not np.allclose(rn.history[:20], r0.history[:20], rtol=1e-6)
