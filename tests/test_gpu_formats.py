"""matrixFormat Coo | Csr | Ell (CsrMatrixWrapper.H:138-161): every format must give the same bits
(the reference executor sums a row in stored order whatever the storage)."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def cfg(fmt, **kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-11, rel_tol=0.0,
                max_iter=300, export_res=1, matrix_format=fmt, adapt_min_iter=0)
    base.update(kw)
    return capi.default_config(**base)


CASES = [dict(gx=5, gy=4, gz=3), dict(gx=6, gy=5, gz=4, periodic_x=True), dict(gx=33, gy=31, gz=29),
         dict(gx=1, gy=1, gz=1), dict(gx=1031, gy=1, gz=1)]


@pytest.mark.parametrize("kw", CASES, ids=[str(i) for i in range(len(CASES))])
@pytest.mark.parametrize("sym", [True, False])
def test_spmv_same_bits_in_every_format(reg, oracle, kw, sym):
    case = synthetic.poisson_block(symmetric=sym, off_upper=-0.9, off_lower=-0.9 if sym else -1.1, **kw)
    rng = np.random.default_rng(20241016)
    x = rng.uniform(-1, 1, case.n_cells)
    rp, cols, vals = oracle_csr(oracle, case)
    ref = oracle.spmv(rp, cols, vals, x)
    for name, fmt in (("coo", capi.FORMAT_COO), ("csr", capi.FORMAT_CSR), ("ell", capi.FORMAT_ELL)):
        s = reg.solver(f"fmt_{name}", cfg(fmt)).set_matrix(case)
        np.testing.assert_array_equal(s.spmv(x), ref)


def test_ell_with_ragged_rows(reg, oracle):
    # a hub cell makes one row 300 wide: width = 300, almost all slots are padding
    n = 300
    lower = np.zeros(n - 1, np.int32)
    upper = np.arange(1, n, dtype=np.int32)
    rng = np.random.default_rng(3)
    case = synthetic.LduCase(n, lower, upper, rng.uniform(1, 2, n) + n, rng.uniform(-1, 1, n - 1),
                             rng.uniform(-1, 1, n - 1))
    s = reg.solver("ell_hub", cfg(capi.FORMAT_ELL)).set_matrix(case)
    rp, cols, vals = oracle_csr(oracle, case)
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


@pytest.mark.parametrize("solver", [capi.SOLVER_CG, capi.SOLVER_BICGSTAB, capi.SOLVER_GMRES])
def test_solvers_on_ell_match_oracle(reg, oracle, solver):
    sym = solver == capi.SOLVER_CG
    case = synthetic.poisson_case(12, symmetric=sym)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    s = reg.solver(f"ell_solver{solver}", cfg(capi.FORMAT_ELL, solver=solver, krylov_dim=20)).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=300)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        if solver == capi.SOLVER_CG:
            ref = oracle.cg(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), **kw)
        elif solver == capi.SOLVER_BICGSTAB:
            ref = oracle.bicgstab(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), **kw)
        else:
            ref = oracle.gmres(A, b, np.zeros_like(b), oracle.Precond(rp, cols, vals, 1), krylov_dim=20, **kw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x)


def test_ell_values_follow_coefficient_updates(reg, oracle):
    case = synthetic.poisson_case(7)
    s = reg.solver("ell_upd", cfg(capi.FORMAT_ELL)).set_matrix(case)
    x = np.linspace(-1, 1, case.n_cells)
    y1 = s.spmv(x)
    case2 = synthetic.poisson_case(7)
    case2.diag = case.diag * 3.0
    s.set_matrix(case2)
    rp, cols, vals = oracle_csr(oracle, case2)
    y2 = s.spmv(x)
    np.testing.assert_array_equal(y2, oracle.spmv(rp, cols, vals, x))
    assert not np.array_equal(y1, y2)
