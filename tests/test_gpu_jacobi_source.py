"""Scalar Jacobi generated from the diagonal as the lduMatrix holds it (the staged source of the coefficient update,
k_jacobi_generate_diag) instead of a strided walk over the CSR values (k_jacobi_generate_pos): the same 1 / d, so the
solves carry the same bits -- and the path steps aside where the device rows are not the caller's cells (renumbered
copy), where a same-rank interface adds entries, and on the reorderOnHost path, which never fills the staged source."""
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


def solve(reg, name, case, b, source, **kw):
    kw.setdefault("solver", capi.SOLVER_CG)
    cfg = capi.default_config(preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0, update_init_guess=1,
                              tolerance=1e-9, rel_tol=0.0, max_iter=300, **kw)
    s = reg.solver(name, cfg)
    s.set_property("jacobiFromSource", source)
    s.set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    return s, x, perf.n_iterations, s.history().copy()


@pytest.mark.parametrize("asym", [False, True])
@pytest.mark.parametrize("n", [20, 48])
def test_same_bits_as_the_walk_over_the_csr_values_and_the_oracle(reg, oracle, n, asym):
    case = synthetic.poisson_case(n, symmetric=not asym)
    b = synthetic.rhs_for_x_star(case)[0]
    kw = dict(solver=capi.SOLVER_BICGSTAB) if asym else {}
    got = {}
    for source in (1.0, 0.0):
        cfg_kw = dict(kw)
        s, x, it, h = solve(reg, f"jsrc_{n}_{asym}_{source}", case, b, source, **cfg_kw)
        assert s.get_property("jacobiFromSourceInUse") == source
        got[source] = (x, it, h)
    assert got[1.0][1] == got[0.0][1]
    np.testing.assert_array_equal(got[1.0][2], got[0.0][2])
    np.testing.assert_array_equal(got[1.0][0], got[0.0][0])
    if not asym:
        A, (rp, cols, vals) = oracle_matrix(oracle, case)
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.cg(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=1e-9, rel_tol=0.0,
                            max_iter=300)
        assert ref.n_iterations == got[1.0][1]
        np.testing.assert_array_equal(got[1.0][2], ref.history)
        np.testing.assert_array_equal(got[1.0][0], ref.x)


def test_new_coefficients_give_a_new_diagonal(reg):
    """Values-only refresh: the preconditioner of the second solve is the one of the second matrix."""
    case = synthetic.poisson_case(24)
    b = synthetic.rhs_for_x_star(case)[0]
    out = {}
    for source in (1.0, 0.0):
        s, _, _, _ = solve(reg, f"jsrc_refresh_{source}", case, b, source)
        case2 = synthetic.poisson_case(24)
        case2.diag[:] = case.diag * 1.5
        s.set_matrix(case2)
        x, perf = s.solve(b, np.zeros_like(b))
        out[source] = (x, perf.n_iterations, s.history().copy())
    assert out[1.0][1] == out[0.0][1]
    np.testing.assert_array_equal(out[1.0][2], out[0.0][2])
    np.testing.assert_array_equal(out[1.0][0], out[0.0][0])


def test_steps_aside(reg):
    """Renumbered device copy, reorderOnHost, a cyclic (same-rank) interface: the walk over the CSR values stays."""
    case = synthetic.renumber_case(synthetic.poisson_case(16), 4096)
    b = synthetic.rhs_for_x_star(case)[0]
    s, _, _, _ = solve(reg, "jsrc_renumbered", case, b, 1.0, renumber=capi.RENUMBER_ON)
    assert s.get_property("jacobiFromSourceInUse") == 0.0
    case = synthetic.poisson_case(16)
    s, _, _, _ = solve(reg, "jsrc_hostorder", case, synthetic.rhs_for_x_star(case)[0], 1.0, reorder_on_host=1)
    assert s.get_property("jacobiFromSourceInUse") == 0.0
    case = synthetic.poisson_block(12, 12, 12, periodic_x=True)
    s, _, _, _ = solve(reg, "jsrc_cyclic", case, synthetic.rhs_for_x_star(case)[0], 1.0)
    assert s.get_property("jacobiFromSourceInUse") == 0.0
