"""Property test (hypothesis) on the GPU: random small lduMatrix systems (irregular face sets, cyclic
patch pairs, symmetric or not) -- device coefficients, SpMV and the solver histories must be
bit-identical to the oracle run in the device's reduction order."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix, oracle_matrix_renumbered, oracle_precond_renumbered, to_new

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


COMBOS = st.sampled_from([
    # (solver, preconditioner, maxBlockSize)   -- CG / ISAI only on the symmetric systems
    ("cg", "bj", 1), ("cg", "none", 1), ("cg", "bj", 3), ("cg", "bj", 6), ("cg", "isai", 1),
    ("bicgstab", "bj", 1), ("bicgstab", "none", 1), ("bicgstab", "bj", 4), ("bicgstab", "gisai", 1),
    ("gmres", "bj", 1), ("gmres", "none", 1), ("gmres", "bj", 5), ("gmres", "gisai", 1),
])


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture,
                                                                 HealthCheck.too_slow])
@given(systems(), COMBOS, st.integers(3, 12), st.booleans())
def test_random_systems_bit_identical(reg, oracle, sysdata, combo, krylov_dim, renumber):
    case, x, b = sysdata
    sym = case.lower is None
    solver, precond, k = combo
    if not sym and solver == "cg":
        solver = "bicgstab"
    if not sym and precond == "isai":
        precond = "gisai"
    kw = dict(tolerance=1e-12, rel_tol=0.0, max_iter=60)
    cfg = capi.default_config(
        solver={"cg": capi.SOLVER_CG, "bicgstab": capi.SOLVER_BICGSTAB, "gmres": capi.SOLVER_GMRES}[solver],
        preconditioner={"bj": capi.PRECOND_BJ, "none": capi.PRECOND_NONE, "isai": capi.PRECOND_ISAI,
                        "gisai": capi.PRECOND_GISAI}[precond],
        max_block_size=k, krylov_dim=krylov_dim, export_res=1, adapt_min_iter=0, update_init_guess=1,
        renumber=capi.RENUMBER_ON if renumber else capi.RENUMBER_OFF, **kw)
    s = reg.solver(f"rand_{solver}_{precond}_{k}_{int(sym)}", cfg).set_matrix(case)
    new_id = s.renumbering()
    assert (new_id is not None) == renumber
    if renumber:
        # the oracle solves the system in the numbering the product chose (explicit input); vectors
        # cross the C ABI in the caller's order
        A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
        d_rp, d_cols, _, d_vals = s.local_matrix()
        np.testing.assert_array_equal(d_rp, rp)
        np.testing.assert_array_equal(d_cols, cols)
        np.testing.assert_array_equal(d_vals, vals)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, to_new(x, new_id))[new_id])
        b_o = to_new(b, new_id)
    else:
        A, (rp, cols, vals) = oracle_matrix(oracle, case)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
        b_o = b
    # ISAI rows of up to 64 pattern entries (<= 32: one thread per row, wider: one wavefront per row);
    # the product refuses wider ones, the oracle too
    P, too_wide = None, False
    # (a renumbered device copy keeps the preconditioner of the caller's numbering: blocks / triangle through new_id)
    mk = (lambda *a, **kw_: oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, *a, **kw_)) if renumber \
        else (lambda *a, **kw_: oracle.Precond(rp, cols, vals, *a, **kw_))
    if precond == "bj":
        P = mk(k)
    elif precond != "none":
        try:
            P = mk(isai="spd" if precond == "isai" else "general")
        except ValueError:
            too_wide = True
    try:
        xs, perf = s.solve(b, np.zeros_like(b))
    except capi.OglError as e:
        assert too_wide and e.status == capi.ERR_UNSUPPORTED, e
        return
    assert not too_wide
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        if solver == "cg":
            ref = oracle.cg(A, b_o, np.zeros_like(b), P, **kw)
        elif solver == "bicgstab":
            ref = oracle.bicgstab(A, b_o, np.zeros_like(b), P, **kw)
        else:
            ref = oracle.gmres(A, b_o, np.zeros_like(b), P, krylov_dim=krylov_dim, **kw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(xs, ref.x[new_id] if renumber else ref.x)
