"""GPU tests of the half storage of a symmetric matrix (config symmetric_half, k_spmv_sym): the device keeps
diagonal + upper coefficients only and reads A(r, r - d) where it reads A(r - d, r).  Rows are summed in
ascending column order, so everything must be bit-identical to full storage and to the oracle."""
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


def cfg(half, **kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-11, rel_tol=0.0,
                max_iter=300, export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0,
                compress_indices=1, symmetric_half=half)
    base.update(kw)
    return capi.default_config(**base)


CASES = [dict(gx=5, gy=4, gz=3), dict(gx=33, gy=31, gz=29), dict(gx=1, gy=1, gz=1), dict(gx=1031, gy=1, gz=1),
         dict(gx=64, gy=64, gz=1), dict(gx=64, gy=64, gz=3), dict(gx=16, gy=16, gz=16),
         dict(gx=37, gy=5, gz=41)]


def randomise(case, seed):
    """Different coefficient on every face (the same for both directions: the matrix stays symmetric)."""
    rng = np.random.default_rng(seed)
    case.upper[:] = rng.uniform(-1.0, -0.25, case.upper.size)
    case.diag[:] = rng.uniform(7.0, 9.0, case.n_cells)
    return case


@pytest.mark.parametrize("kw", CASES, ids=[str(i) for i in range(len(CASES))])
def test_spmv_same_bits_half_or_full_storage(reg, oracle, kw):
    case = randomise(synthetic.poisson_block(**kw), 3)
    rng = np.random.default_rng(20241016)
    x = rng.uniform(-1, 1, case.n_cells)
    rp, cols, vals = oracle_csr(oracle, case)
    ref = oracle.spmv(rp, cols, vals, x)
    half = reg.solver("sym_half", cfg(1)).set_matrix(case)
    full = reg.solver("sym_full", cfg(0)).set_matrix(case)
    # (one set of distances for the whole matrix, else the per-chunk variant with explicit exceptions)
    qualifies = capi.host_sym_check(rp, cols)[0] or capi.host_symx_check(rp, cols)[0]
    assert half.get_property("symmetricHalf") == (1.0 if qualifies else 0.0)
    assert half.get_property("symmetricHalfPerChunk") == (0.0 if capi.host_sym_check(rp, cols)[0] or not qualifies else 1.0)
    assert full.get_property("symmetricHalf") == 0.0
    assert half.get_property("spmvLayout") == full.get_property("spmvLayout") == 2.0
    np.testing.assert_array_equal(half.spmv(x), ref)
    np.testing.assert_array_equal(full.spmv(x), ref)
    if case.n_cells > 1:
        assert qualifies


@pytest.mark.parametrize("solver", [capi.SOLVER_CG, capi.SOLVER_BICGSTAB, capi.SOLVER_GMRES])
@pytest.mark.parametrize("precond", [capi.PRECOND_BJ, capi.PRECOND_NONE])
def test_solvers_same_history_as_the_oracle(reg, oracle, solver, precond):
    case = randomise(synthetic.poisson_case(20), 5)          # 8000 rows: 16 chunks, the last one partial
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    out = {}
    for half in (1, 0):
        s = reg.solver(f"sym_solver{solver}_{precond}_{half}",
                       cfg(half, solver=solver, preconditioner=precond, krylov_dim=20)).set_matrix(case)
        assert s.get_property("symmetricHalf") == float(half)
        x, perf = s.solve(b, np.zeros_like(b))
        out[half] = (x, s.history().copy(), perf.n_iterations)
    np.testing.assert_array_equal(out[1][1], out[0][1])
    np.testing.assert_array_equal(out[1][0], out[0][0])
    assert out[1][2] == out[0][2]
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=300)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        if solver == capi.SOLVER_GMRES:
            P = oracle.Precond(rp, cols, vals, 1) if precond else None
            ref = oracle.gmres(A, b, np.zeros_like(b), P, krylov_dim=20, **kw)
        else:
            fn = oracle.cg if solver == capi.SOLVER_CG else oracle.bicgstab
            ref = fn(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals) if precond else None, **kw)
    np.testing.assert_array_equal(out[1][1], ref.history)
    np.testing.assert_array_equal(out[1][0], ref.x)


def test_planes_follow_coefficient_updates_and_pattern_changes(reg, oracle):
    s = reg.solver("sym_upd", cfg(1))
    rng = np.random.default_rng(9)
    for n, seed in ((9, 1), (9, 2), (12, 3), (9, 4)):
        case = randomise(synthetic.poisson_case(n), seed)
        s.set_matrix(case)
        assert s.get_property("symmetricHalf") == 1.0
        rp, cols, vals = oracle_csr(oracle, case)
        x = rng.uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


def test_matrices_that_keep_full_storage(reg, oracle):
    rng = np.random.default_rng(2)
    cases = {"asymmetric": synthetic.poisson_case(12, symmetric=False),
             "cyclic pair": synthetic.poisson_block(6, 5, 4, periodic_x=True),
             "shuffled": synthetic.renumber_case(synthetic.poisson_case(12), 256)}
    for name, case in cases.items():
        s = reg.solver("sym_" + name.replace(" ", "_"), cfg(1)).set_matrix(case)
        assert s.get_property("symmetricHalf") == 0.0, name
        rp, cols, vals = oracle_csr(oracle, case)
        x = rng.uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
    # a forced renumbering (RCM) leaves no bands: full storage in the new numbering
    case = synthetic.poisson_case(30)
    s = reg.solver("sym_rcm", cfg(1, renumber=capi.RENUMBER_ON)).set_matrix(case)
    assert s.renumbering() is not None and s.get_property("symmetricHalf") == 0.0
    # toggling the keyword on the same field rebuilds the device copy
    s = reg.solver("sym_toggle", cfg(1)).set_matrix(synthetic.poisson_case(10))
    assert s.get_property("symmetricHalf") == 1.0
    s2 = reg.solver("sym_toggle", cfg(0)).set_matrix(synthetic.poisson_case(10))
    assert s2.get_property("symmetricHalf") == 0.0


def test_random_boxes_same_bits_as_the_oracle(reg, oracle):
    """Boxes of random shape (even and odd line lengths -> both instantiations of the kernel, 1-D / 2-D / 3-D,
    partial last chunks) with random symmetric coefficients: SpMV, residual SpMV (first CG residual) and a
    short CG history against the oracle."""
    rng = np.random.default_rng(20241016)
    shapes = [(int(rng.integers(1, 70)), int(rng.integers(1, 40)), int(rng.integers(1, 30))) for _ in range(24)]
    shapes += [(64, 32, 16), (2, 2, 2), (128, 1, 1), (1, 96, 1), (1, 1, 77), (66, 1, 34)]
    for i, (gx, gy, gz) in enumerate(shapes):
        case = randomise(synthetic.poisson_block(gx=gx, gy=gy, gz=gz), 100 + i)
        rp, cols, vals = oracle_csr(oracle, case)
        s = reg.solver("sym_rand", cfg(1, max_iter=12)).set_matrix(case)
        qualifies = capi.host_sym_check(rp, cols)[0] or capi.host_symx_check(rp, cols)[0]
        assert s.get_property("symmetricHalf") == (1.0 if qualifies else 0.0), (gx, gy, gz)
        x = rng.uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x), err_msg=str((gx, gy, gz)))
        b = rng.uniform(-1, 1, case.n_cells)
        xs, perf = s.solve(b, x.copy())
        A, _ = oracle_matrix(oracle, case)
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = oracle.cg(A, b, x.copy(), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=1e-11, rel_tol=0.0,
                            max_iter=12)
        np.testing.assert_array_equal(s.history(), ref.history, err_msg=str((gx, gy, gz)))
        np.testing.assert_array_equal(xs, ref.x, err_msg=str((gx, gy, gz)))


def test_field_switching_between_symmetric_and_asymmetric_matrices(reg, oracle):
    """The same field handed a symmetric, then an asymmetric, then again a symmetric matrix on the same
    addressing: the half storage must come and go with the `lower` pointer."""
    s = reg.solver("sym_switch", cfg(1))
    rng = np.random.default_rng(31)
    for symmetric, expect in ((True, 1.0), (False, 0.0), (True, 1.0), (True, 1.0), (False, 0.0)):
        case = synthetic.poisson_case(14, symmetric=symmetric)
        case.upper[:] = rng.uniform(-1.0, -0.25, case.upper.size)
        if not symmetric:
            case.lower[:] = rng.uniform(-1.0, -0.25, case.lower.size)
        s.set_matrix(case)
        assert s.get_property("symmetricHalf") == expect
        rp, cols, vals = oracle_csr(oracle, case)
        x = rng.uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


# ---- per-chunk distances + explicit exceptions (k_spmv_symx): symmetric matrices that are banded only locally ----
SYMX_MESHES = [
    ("two_blocks", lambda: synthetic.multi_block_case([30, 17], 24, 20)),
    ("three_blocks", lambda: synthetic.multi_block_case([12, 40, 9], 16, 33)),
    ("odd_blocks", lambda: synthetic.multi_block_case([21, 33], 17, 13)),
    ("long_blocks", lambda: synthetic.multi_block_case([300, 260], 5, 7)),
]


@pytest.mark.parametrize("name,make", SYMX_MESHES, ids=[m[0] for m in SYMX_MESHES])
@pytest.mark.parametrize("stream", [1e18, 0.0], ids=["cached", "streamed"])
def test_multi_block_meshes_run_on_per_chunk_half_storage(reg, oracle, name, make, stream):
    """A multi-block structured mesh is banded block by block; the couplings across the block faces sit at
    distances of their own.  The half storage takes its distances per chunk and keeps what does not fit as
    explicit entries merged into the row sum by column: SpMV, residual SpMV (first CG residual), fused dots and
    whole histories bit-identical to the oracle and to full storage, for every solver."""
    case = randomise(make(), 21)
    rp, cols, vals = oracle_csr(oracle, case)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, case.n_cells)
    b = rng.uniform(-1, 1, case.n_cells)
    A, _ = oracle_matrix(oracle, case)
    inv = oracle.jacobi_generate_scalar(rp, cols, vals)
    for solver, fn in ((capi.SOLVER_CG, oracle.cg), (capi.SOLVER_BICGSTAB, oracle.bicgstab)):
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = fn(A, b, x.copy(), inv, tolerance=1e-11, rel_tol=0.0, max_iter=25)
        for half in (1, 0):
            s = reg.solver(f"symx_{name}_{solver}_{half}_{int(stream > 0)}",
                           cfg(half, solver=solver, max_iter=25, update_init_guess=1))
            s.set_property("streamAboveBytes", stream)
            s.set_matrix(case)
            assert s.get_property("symmetricHalf") == float(half)
            assert s.get_property("symmetricHalfPerChunk") == float(half)
            assert s.get_property("spmvLayout") == 2.0
            np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
            xs, perf = s.solve(b, x.copy())
            np.testing.assert_array_equal(s.history(), ref.history, err_msg=f"{name} solver {solver} half {half}")
            np.testing.assert_array_equal(xs, ref.x)
    P = oracle.Precond(rp, cols, vals, 1)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.gmres(A, b, x.copy(), P, krylov_dim=10, tolerance=1e-11, rel_tol=0.0, max_iter=25)
    s = reg.solver(f"symx_{name}_gmres_{int(stream > 0)}",
                   cfg(1, solver=capi.SOLVER_GMRES, krylov_dim=10, max_iter=25, update_init_guess=1))
    s.set_property("streamAboveBytes", stream)
    s.set_matrix(case)
    xs, perf = s.solve(b, x.copy())
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(xs, ref.x)


def test_per_chunk_half_storage_follows_updates_and_pattern_changes(reg, oracle):
    s = reg.solver("symx_upd", cfg(1))
    rng = np.random.default_rng(9)
    seq = [(lambda: synthetic.multi_block_case([30, 17], 24, 20), 1.0, 1.0), (lambda: synthetic.multi_block_case([30, 17], 24, 20), 1.0, 1.0),
           (lambda: synthetic.poisson_case(20), 1.0, 0.0), (lambda: synthetic.multi_block_case([12, 40, 9], 16, 33), 1.0, 1.0),
           (lambda: synthetic.renumber_case(synthetic.poisson_case(16), 512), 0.0, 0.0)]
    for i, (make, half, per_chunk) in enumerate(seq):
        case = randomise(make(), 30 + i)
        s.set_matrix(case)
        assert s.get_property("symmetricHalf") == half and s.get_property("symmetricHalfPerChunk") == per_chunk
        rp, cols, vals = oracle_csr(oracle, case)
        x = rng.uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


def with_extra_faces(case, first_row, last_row, count, seed):
    """`count` more faces between random cells of [first_row, last_row): couplings no plane holds."""
    rng = np.random.default_rng(seed)
    have = set(zip(case.lower_addr.tolist(), case.upper_addr.tolist()))
    lo, up = [], []
    while len(lo) < count:
        a, b = sorted(int(v) for v in rng.integers(first_row, last_row, 2))
        if a != b and (a, b) not in have:
            have.add((a, b))
            lo.append(a)
            up.append(b)
    la = np.concatenate([case.lower_addr, np.array(lo, np.int32)])
    ua = np.concatenate([case.upper_addr, np.array(up, np.int32)])
    coef = np.concatenate([case.upper, rng.uniform(-1.0, -0.25, count)])
    order = np.lexsort((ua, la))
    diag = case.diag.copy()
    np.add.at(diag, np.array(lo), 1.0)
    np.add.at(diag, np.array(up), 1.0)
    return synthetic.LduCase(case.n_cells, la[order].astype(np.int32), ua[order].astype(np.int32), diag, coef[order], None,
                             [], case.global_index, case.global_n)


SYMX_IRREGULAR = [
    # (name, case, chunks the general kernel takes: few (<= 2) / some)
    ("random_band_odd_n", lambda: synthetic.random_global_case(1029, 3, 4, seed=8), "some"),
    ("random_band_even_n", lambda: synthetic.random_global_case(2050, 2, 4, seed=3), "some"),
    ("blocks_plus_scattered_faces", lambda: with_extra_faces(synthetic.multi_block_case([30, 17], 24, 20), 0, 3000, 400, 4), "some"),
    # 1400 explicit entries in the first chunk: more than the general kernel stages through LDS (it reads them from memory)
    ("blocks_plus_dense_patch", lambda: with_extra_faces(synthetic.multi_block_case([30, 17], 24, 20), 0, 512, 700, 5), "some"),
    ("blocks_only", lambda: synthetic.multi_block_case([30, 17], 24, 20), "few"),   # (the chunk that straddles the two blocks)
]


@pytest.mark.parametrize("name,make,general", SYMX_IRREGULAR, ids=[m[0] for m in SYMX_IRREGULAR])
def test_per_chunk_half_storage_lean_and_general_chunks(reg, oracle, name, make, general):
    """The lean kernel takes the chunks whose rows have at most one explicit entry ahead of and one behind their
    planar ones, the general kernel (a second launch) the others; systems with an odd number of rows end in a pair
    load at the last even row.  SpMV, residual SpMV and a CG history as the oracle's, streamed and cached."""
    case = make()
    if name.startswith("blocks"):
        case = randomise(case, 23) if name == "blocks_only" else case
    ok, _, planar, explicit, _, chunks = capi.host_symx_check(*oracle_csr(oracle, case)[:2])
    if not ok:
        pytest.skip("layout does not qualify")
    rp, cols, vals = oracle_csr(oracle, case)
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, case.n_cells)
    b = rng.uniform(-1, 1, case.n_cells)
    A, _ = oracle_matrix(oracle, case)
    inv = oracle.jacobi_generate_scalar(rp, cols, vals)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, b, x.copy(), inv, tolerance=1e-11, rel_tol=0.0, max_iter=25)
    for stream in (1e18, 0.0):
        s = reg.solver(f"symx_irr_{name}_{int(stream > 0)}", cfg(1, max_iter=25, update_init_guess=1))
        s.set_property("streamAboveBytes", stream)
        s.set_matrix(case)
        assert s.get_property("symmetricHalfPerChunk") == 1.0
        g = s.get_property("symxGeneralChunks")
        assert {"few": g <= 2, "some": 0 < g}[general], (g, chunks)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
        xs, perf = s.solve(b, x.copy())
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(xs, ref.x)
