"""BASELINE.json's full size (216^3 = 10,077,696 rows, 70,263,936 nnz) on the MI355X, checked
through size-independent properties (the oracle would need minutes per case at this size, so it
only spot-checks a window of rows):

  * A.1 = delta        row sums of the synthetic Poisson matrix are its diagonal shift
  * linearity          A(ax + by) = a Ax + b Ay   (to rounding)
  * symmetry           x.(A y) = y.(A x)          (device dot, to rounding)
  * spot rows          a 512-row chunk window of y = A x is bit-identical to the oracle's SpMV
  * round trip         solve A x = A x* and recover x*; the reported normalised-L1 residual equals
                       the one recomputed from the returned x; CG and BiCGStab agree
  * device pattern     ldu_mapping is a permutation-with-repeats onto [upper|diag]; coefficients
                       are a checksum-preserving gather of the LDU arrays
"""
import numpy as np
import pytest

from ogl_amd import capi, synthetic

pytestmark = pytest.mark.gpu
N_EDGE = 216


@pytest.fixture(scope="module")
def big():
    case = synthetic.poisson_case(N_EDGE)
    reg = capi.Registry()
    cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-9,
                              rel_tol=0.0, max_iter=2000, export_res=1, adapt_min_iter=0,
                              matrix_format=capi.FORMAT_CSR)
    s = reg.solver("p", cfg).set_matrix(case)
    yield case, reg, s
    reg.close()


def test_row_sums_are_the_diagonal_shift(big):
    case, reg, s = big
    y = s.spmv(np.ones(case.n_cells))
    delta = 1e-3 * (1.0 + (case.global_index % 7) / 7.0)
    np.testing.assert_allclose(y, delta, rtol=0, atol=4e-15)


def test_linearity_and_symmetry(big):
    case, reg, s = big
    rng = np.random.default_rng(20241016)
    x, y = rng.uniform(-1, 1, case.n_cells), rng.uniform(-1, 1, case.n_cells)
    ax, ay = s.spmv(x), s.spmv(y)
    lin = s.spmv(0.5 * x - 2.0 * y)
    np.testing.assert_allclose(lin, 0.5 * ax - 2.0 * ay, rtol=0, atol=1e-13)
    xay, yax = s.reduce("dot", x, ay), s.reduce("dot", y, ax)
    assert xay == pytest.approx(yax, rel=1e-11)
    # deterministic: same launch, same bits
    np.testing.assert_array_equal(s.spmv(x), ax)
    assert s.reduce("dot", x, ay) == xay


def test_spot_rows_bit_identical_to_oracle(big, oracle):
    case, reg, s = big
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, case.n_cells)
    y = s.spmv(x)
    rp, cols, mp, vals = s.local_matrix()
    for r0 in (0, 512 * 9841, case.n_cells - 512):          # first, a middle and the last chunk
        sub_rp = (rp[r0:r0 + 513] - rp[r0]).astype(np.int32)
        lo, hi = rp[r0], rp[r0 + 512]
        ref = oracle.spmv(sub_rp, cols[lo:hi], vals[lo:hi], x)
        np.testing.assert_array_equal(y[r0:r0 + 512], ref)


def test_device_pattern_and_coefficients(big):
    case, reg, s = big
    rp, cols, mp, vals = s.local_matrix()
    N, F = case.n_cells, case.n_faces
    assert rp[0] == 0 and rp[-1] == N + 2 * F
    assert np.all(np.diff(rp) >= 4) and np.all(np.diff(rp) <= 7)
    # strictly increasing columns inside rows (sortedness, data_validation.py's row-major check)
    inner = np.ones(cols.size, bool)
    inner[rp[1:-1]] = False
    assert np.all(np.diff(cols)[inner[1:]] > 0)
    # ldu_mapping: every face twice (upper + its transpose), every diagonal once
    counts = np.bincount(mp, minlength=F + N)
    assert np.all(counts[:F] == 2) and np.all(counts[F:] == 1)
    # coefficient gather preserves the checksum: sum(vals) = 2 sum(upper) + sum(diag)
    # (float sums of 70M entries of mixed sign: compare to 1e-9, the gather itself is checked bitwise below)
    assert vals.sum() == pytest.approx(2 * case.upper.sum() + case.diag.sum(), rel=1e-9)
    src = np.concatenate([case.upper, case.diag])
    np.testing.assert_array_equal(vals, src[mp])
    # sign pattern of a Poisson matrix (data_validation.py:93-111 analogue)
    diag_pos = cols == np.repeat(np.arange(N, dtype=np.int32), np.diff(rp))
    assert np.all(vals[diag_pos] > 0) and np.all(vals[~diag_pos] < 0)


def test_round_trip_cg_and_bicgstab(big):
    case, reg, s = big
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = s.spmv(xs)
    s.upload_solution(None)
    x, perf = s.solve(b, np.zeros_like(b))
    hist = s.history()
    assert perf.final_residual < 1e-9 and hist.size == perf.n_iterations
    assert hist[0] == perf.initial_residual and hist[-1] == perf.final_residual
    np.testing.assert_allclose(x, xs, rtol=0, atol=1e-6)
    # the reported residual is the normalised L1 residual of the returned x
    r = b - s.spmv(x)
    assert np.abs(r).sum() / perf.norm_factor == pytest.approx(perf.final_residual, rel=1e-3)
    xbar = np.full_like(b, 0.0)
    nf = np.abs(s.spmv(xbar) - s.spmv(xbar)).sum() + np.abs(b - s.spmv(xbar)).sum() + 1e-15
    assert perf.norm_factor == pytest.approx(nf, rel=1e-12)
    # same answer through GKOBiCGStab on the same persistent matrix (second field)
    cfg = capi.default_config(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_BJ,
                              tolerance=1e-9, rel_tol=0.0, max_iter=2000, adapt_min_iter=0,
                              matrix_format=capi.FORMAT_CSR)
    s2 = reg.solver("p2", cfg).set_matrix(case)
    x2, perf2 = s2.solve(b, np.zeros_like(b))
    assert perf2.final_residual < 1e-9
    np.testing.assert_allclose(x2, x, rtol=0, atol=1e-6)
