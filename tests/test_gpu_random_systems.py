"""Property test (hypothesis) on the GPU: random small lduMatrix systems (irregular face sets, cyclic
patch pairs, symmetric or not) -- device coefficients, SpMV and the solver histories must be
bit-identical to the oracle run in the device's reduction order."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


@st.composite
def systems(draw):
    n = draw(st.integers(2, 700))                       # one or two chunks of 512 rows
    seed = draw(st.integers(0, 2 ** 31))
    rng = np.random.default_rng(seed)
    per_row = draw(st.integers(1, 5))
    pairs = set()
    for i in range(n - 1):
        for j in rng.integers(i + 1, min(n, i + 1 + draw(st.sampled_from([3, 40, 1000]))), per_row):
            pairs.add((i, int(j)))
    pairs = np.array(sorted(pairs), dtype=np.int32).reshape(-1, 2)
    f = len(pairs)
    sym = draw(st.booleans())
    upper = rng.uniform(-1, 0, f)
    lower = None if sym else rng.uniform(-1, 0, f)
    ifaces = []
    if draw(st.booleans()):
        m = draw(st.integers(1, min(8, n)))
        a = rng.choice(n, m, replace=False).astype(np.int32)
        b = rng.choice(n, m, replace=False).astype(np.int32)
        c = rng.uniform(0, 0.5, m)
        ifaces = [synthetic.Interface(synthetic.IFACE_CYCLIC, a, c, -1, 1),
                  synthetic.Interface(synthetic.IFACE_CYCLIC, b, c if sym else rng.uniform(0, 0.5, m), -1, 0)]
    # diagonally dominant: CG / BiCGStab converge
    diag = np.full(n, 1.0)
    np.add.at(diag, pairs[:, 0], np.abs(upper))
    np.add.at(diag, pairs[:, 1], np.abs(upper if sym else lower))
    for itf in ifaces:
        np.add.at(diag, itf.face_cells, np.abs(itf.bou_coeffs))
    case = synthetic.LduCase(n, pairs[:, 0].copy(), pairs[:, 1].copy(), diag, upper, lower, ifaces)
    return case, rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture,
                                                                 HealthCheck.too_slow])
@given(systems())
def test_random_systems_bit_identical(reg, oracle, sysdata):
    case, x, b = sysdata
    sym = case.lower is None
    solver = capi.SOLVER_CG if sym else capi.SOLVER_BICGSTAB
    kw = dict(tolerance=1e-12, rel_tol=0.0, max_iter=60)
    cfg = capi.default_config(solver=solver, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
                              update_init_guess=1, **kw)
    s = reg.solver("rand_sym" if sym else "rand_asym", cfg).set_matrix(case)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    rp_d, cols_d, vals_d = s.local_matrix_csr() if hasattr(s, "local_matrix_csr") else (None, None, None)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
    xs, perf = s.solve(b, np.zeros_like(b))
    inv = oracle.jacobi_generate_scalar(rp, cols, vals)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = (oracle.cg if sym else oracle.bicgstab)(A, b, np.zeros_like(b), inv, **kw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(xs, ref.x)
