"""compress_indices: the Coo/Csr formats run the SpMV on the index-compressed chunked ELL copy when the
pattern qualifies.  Same bits as the CSR-stream kernel and as the oracle, everywhere it is used; the
CSR-stream kernel takes over when the pattern does not qualify."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix

pytestmark = pytest.mark.gpu

LAYOUT_CSR, LAYOUT_ELL, LAYOUT_SELL = 0.0, 1.0, 2.0


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def cfg(compress, **kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-11, rel_tol=0.0,
                max_iter=300, export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0,
                compress_indices=compress)
    base.update(kw)
    return capi.default_config(**base)


CASES = [dict(gx=5, gy=4, gz=3), dict(gx=6, gy=5, gz=4, periodic_x=True), dict(gx=33, gy=31, gz=29),
         dict(gx=1, gy=1, gz=1), dict(gx=1031, gy=1, gz=1), dict(gx=64, gy=64, gz=3)]


@pytest.mark.parametrize("kw", CASES, ids=[str(i) for i in range(len(CASES))])
@pytest.mark.parametrize("sym", [True, False])
def test_spmv_same_bits_compressed_or_not(reg, oracle, kw, sym):
    case = synthetic.poisson_block(symmetric=sym, off_upper=-0.9, off_lower=-0.9 if sym else -1.1, **kw)
    rng = np.random.default_rng(20241016)
    x = rng.uniform(-1, 1, case.n_cells)
    rp, cols, vals = oracle_csr(oracle, case)
    ref = oracle.spmv(rp, cols, vals, x)
    on = reg.solver("sell_on", cfg(1)).set_matrix(case)
    off = reg.solver("sell_off", cfg(0)).set_matrix(case)
    assert on.get_property("spmvLayout") == LAYOUT_SELL
    assert off.get_property("spmvLayout") == LAYOUT_CSR
    np.testing.assert_array_equal(on.spmv(x), ref)
    np.testing.assert_array_equal(off.spmv(x), ref)


def test_default_config_compresses_and_ell_format_does_not(reg):
    case = synthetic.poisson_case(9)
    assert capi.default_config().compress_indices == 1
    s = reg.solver("sell_default", capi.default_config()).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    e = reg.solver("sell_ellfmt", cfg(1, matrix_format=capi.FORMAT_ELL)).set_matrix(case)
    assert e.get_property("spmvLayout") == LAYOUT_ELL


def random_ldu(n, per_row, seed):
    """symmetric-pattern lduMatrix with `per_row` random upper neighbours per cell, faces in OpenFOAM's
    upper-triangular order"""
    rng = np.random.default_rng(seed)
    pairs = set()
    for i in range(n - 1):
        for j in rng.integers(i + 1, n, per_row):
            pairs.add((i, int(j)))
    pairs = np.array(sorted(pairs), dtype=np.int32)
    f = len(pairs)
    return synthetic.LduCase(n, pairs[:, 0].copy(), pairs[:, 1].copy(), rng.uniform(1, 2, n) + 4 * per_row,
                             rng.uniform(-1, 1, f), rng.uniform(-1, 1, f))


def test_unstructured_pattern_runs_on_16_bit_deltas(reg, oracle):
    # a box whose cells were renumbered at random inside windows of 1000: every chunk sees far more than
    # 255 distinct offsets, every distance between two columns of a row fits 16 bits
    case = synthetic.renumber_case(synthetic.poisson_case(20, symmetric=False), 1000)
    s = reg.solver("sell_random", cfg(1, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    rp, cols, vals = oracle_csr(oracle, case)
    ok, d16, c32 = capi.host_sell_modes(rp, cols)
    assert ok and d16 >= 14 and c32 == 0
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, case.n_cells)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
    # same bits from the CSR-stream kernel
    off = reg.solver("sell_random_off", cfg(0, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert off.get_property("spmvLayout") == LAYOUT_CSR
    np.testing.assert_array_equal(off.spmv(x), s.spmv(x))


def test_far_couplings_take_32_bit_columns_in_their_chunk(reg, oracle):
    # 80,000 cells; cells 600..1100 also couple to a cell 70,000 further on: those chunks store plain
    # 32-bit columns, the rest 16-bit deltas; all of it in one launch
    n = 80_000
    rng = np.random.default_rng(4)
    # every cell couples to three later cells at a distance that grows along its block of 512 cells:
    # 512 distinct offsets per leg and chunk (no 1-byte dictionary fits), one incoming edge per leg and
    # cell (row lengths stay uniform)
    pairs = set()
    for j in (1, 2, 3):
        for a in range(n):
            b_ = a + 600 * j + (a % 512) + (a // 512) % 2
            if b_ < n:
                pairs.add((a, b_))
    pairs |= {(a, a + 70_000) for a in range(600, 1100)}
    pairs = np.array(sorted(p for p in pairs if p[0] < p[1]), dtype=np.int32)
    f = len(pairs)
    case = synthetic.LduCase(n, pairs[:, 0].copy(), pairs[:, 1].copy(), rng.uniform(20, 30, n),
                             rng.uniform(-1, 1, f), rng.uniform(-1, 1, f))
    # (compress_indices 2: without the one-off timing against the CSR-stream kernel that decides for
    # irregular patterns of this size)
    s = reg.solver("sell_far", cfg(2, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    rp, cols, vals = oracle_csr(oracle, case)
    ok, d16, c32 = capi.host_sell_modes(rp, cols)
    assert ok and c32 >= 2 and d16 > 100
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


def test_one_long_row_per_chunk_is_spilled(reg, oracle):
    # padding every row to the chunk's longest would cost far more than CSR's indices: the chunk keeps 3 slots
    # per row and the long row's tail is added by the spill pass
    n = 2048
    lower = [r for r in range(n - 1)] + [r for r in range(7, n, 512) for _ in range(2, 200) if r + 199 < n]
    upper = [r + 1 for r in range(n - 1)] + [r + d for r in range(7, n, 512) for d in range(2, 200) if r + 199 < n]
    order = np.lexsort((upper, lower))
    lower, upper = np.array(lower, np.int32)[order], np.array(upper, np.int32)[order]
    rng = np.random.default_rng(3)
    case = synthetic.LduCase(n, lower, upper, rng.uniform(300, 400, n), rng.uniform(-1, 1, len(lower)), None)
    s = reg.solver("sell_longrow", cfg(1, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL and s.get_property("sellSpilledEntries") > 0
    rp, cols, vals = oracle_csr(oracle, case)
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
    # the fused dot partials of the chunks that hold spilled rows are redone: whole solves stay bit-identical
    b = oracle.spmv(rp, cols, vals, x)
    A, _ = oracle_matrix(oracle, case)
    for solver, fn in ((capi.SOLVER_CG, oracle.cg), (capi.SOLVER_BICGSTAB, oracle.bicgstab)):
        sv = reg.solver(f"sell_longrow_{solver}", cfg(1, solver=solver, renumber=capi.RENUMBER_OFF, max_iter=60)).set_matrix(case)
        xs, perf = sv.solve(b, np.zeros_like(b))
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = fn(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=1e-11,
                     rel_tol=0.0, max_iter=60)
        np.testing.assert_array_equal(sv.history(), ref.history)
        np.testing.assert_array_equal(xs, ref.x)


def test_alternating_short_and_long_rows_fall_back_to_csr_stream(reg, oracle):
    # rows of 2 and 10 entries alternate: half of the rows are long, no cap helps and the padding would cost
    # more than CSR's indices; with `renumber off` nothing is sorted either -> the CSR-stream kernel runs
    n = 4096
    lower = [r for r in range(0, n - 8, 2) for d in range(1, 9)]
    upper = [r + d for r in range(0, n - 8, 2) for d in range(1, 9)]
    lower, upper = np.array(lower, np.int32), np.array(upper, np.int32)
    keep = (upper % 2 == 1)                      # even rows couple to odd rows only: odd rows stay short-ish
    lower, upper = lower[keep], upper[keep]
    rng = np.random.default_rng(3)
    case = synthetic.LduCase(n, lower, upper, rng.uniform(30, 40, n), rng.uniform(-1, 1, len(lower)), None)
    s = reg.solver("sell_altrows", cfg(1, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    rp, cols, vals = oracle_csr(oracle, case)
    if not capi.host_sell_check(rp, cols)[0]:
        assert s.get_property("spmvLayout") == LAYOUT_CSR
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


def test_wide_banded_rows(reg, oracle):
    # 41 diagonals -> three 16-byte code words per thread and 6 value batches
    n = 1500
    lower = np.repeat(np.arange(n, dtype=np.int32), 20)
    upper = lower + np.tile(np.arange(1, 21, dtype=np.int32), n)
    keep = upper < n
    lower, upper = lower[keep], upper[keep]
    rng = np.random.default_rng(8)
    case = synthetic.LduCase(n, lower, upper, rng.uniform(40, 50, n), rng.uniform(-1, 1, len(lower)),
                             rng.uniform(-1, 1, len(lower)))
    s = reg.solver("sell_wide", cfg(1)).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    rp, cols, vals = oracle_csr(oracle, case)
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


@pytest.mark.parametrize("solver", [capi.SOLVER_CG, capi.SOLVER_BICGSTAB, capi.SOLVER_GMRES])
def test_solvers_same_history_compressed_or_not(reg, oracle, solver):
    sym = solver == capi.SOLVER_CG
    case = synthetic.poisson_case(20, symmetric=sym)      # 8000 rows: 16 chunks, the last one partial
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    out = {}
    for tag, c in (("on", 1), ("off", 0)):
        s = reg.solver(f"sell_solver{solver}_{tag}", cfg(c, solver=solver, krylov_dim=20)).set_matrix(case)
        assert s.get_property("spmvLayout") == (LAYOUT_SELL if c else LAYOUT_CSR)
        x, perf = s.solve(b, np.zeros_like(b))
        out[tag] = (x, s.history().copy(), perf.n_iterations)
    np.testing.assert_array_equal(out["on"][1], out["off"][1])
    np.testing.assert_array_equal(out["on"][0], out["off"][0])
    assert out["on"][2] == out["off"][2]
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    kw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=300)
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        if solver == capi.SOLVER_CG:
            ref = oracle.cg(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), **kw)
        elif solver == capi.SOLVER_BICGSTAB:
            ref = oracle.bicgstab(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), **kw)
        else:
            ref = oracle.gmres(A, b, np.zeros_like(b), oracle.Precond(rp, cols, vals, 1), krylov_dim=20, **kw)
    np.testing.assert_array_equal(out["on"][1], ref.history)
    np.testing.assert_array_equal(out["on"][0], ref.x)


def test_compressed_values_follow_coefficient_updates(reg, oracle):
    case = synthetic.poisson_case(9)
    s = reg.solver("sell_upd", cfg(1)).set_matrix(case)
    x = np.random.default_rng(4).uniform(-1, 1, case.n_cells)
    y0 = s.spmv(x)
    case2 = synthetic.poisson_case(9)
    case2.diag[:] = case2.diag * 1.5
    case2.upper[:] = case2.upper * 0.5
    s.set_matrix(case2)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    rp, cols, vals = oracle_csr(oracle, case2)
    y1 = s.spmv(x)
    np.testing.assert_array_equal(y1, oracle.spmv(rp, cols, vals, x))
    assert not np.array_equal(y0, y1)


def test_pattern_change_rebuilds_the_layout(reg, oracle):
    s = reg.solver("sell_repattern", cfg(1))
    for n in (6, 11):
        case = synthetic.poisson_case(n)
        s.set_matrix(case)
        assert s.get_property("spmvLayout") == LAYOUT_SELL
        rp, cols, vals = oracle_csr(oracle, case)
        x = np.random.default_rng(n).uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


def test_band_aware_workgroup_order_follows_a_pattern_rebuild(reg, oracle):
    """Round 5: on a banded pattern the workgroups of the compressed kernel take the chunks in the band-aware order (the
    chunks of rows r and r +- band on one XCD).  The order belongs to the pattern: a rebuild with the SAME band and MORE
    rows (80 x 80 x 80 -> 80 x 80 x 100 cells, band 6400) must not run the new matrix on the old order, which would
    leave its last chunks out; the product does not depend on the order (property spmvBandRows 0)."""
    s = reg.solver("sell_band_rebuild", cfg(1, solver=capi.SOLVER_BICGSTAB))
    for gz in (80, 100):
        case = synthetic.poisson_block(80, 80, gz, symmetric=False, off_upper=-0.9, off_lower=-1.1)
        s.set_matrix(case)
        assert s.get_property("spmvLayout") == LAYOUT_SELL
        rp, cols, vals = oracle_csr(oracle, case)
        x = np.random.default_rng(gz).uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
    s0 = reg.solver("sell_band_off", cfg(1, solver=capi.SOLVER_BICGSTAB))
    s0.set_property("spmvBandRows", 0.0)
    s0.set_matrix(case)
    np.testing.assert_array_equal(s0.spmv(x), s.spmv(x))
    b = oracle.spmv(rp, cols, vals, synthetic.x_star(case.global_index, case.global_n))
    xa, pa = s.solve(b, np.zeros_like(b))
    xb, pb = s0.solve(b, np.zeros_like(b))
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(s.history(), s0.history())


def test_offset_mode_when_a_chunk_has_too_many_row_patterns(reg, oracle):
    # every row couples to a pseudo-random 90 % of the 20 following rows: hundreds of distinct row
    # patterns per chunk but only 41 distinct offsets -> the chunks use one byte per (row, slot)
    n = 1300
    rng = np.random.default_rng(12)
    pairs = [(i, i + d) for i in range(n) for d in range(1, 21) if i + d < n and rng.random() < 0.9]
    pairs = np.array(pairs, dtype=np.int32)
    f = len(pairs)
    case = synthetic.LduCase(n, pairs[:, 0].copy(), pairs[:, 1].copy(), rng.uniform(40, 50, n),
                             rng.uniform(-1, 1, f), rng.uniform(-1, 1, f))
    rp, cols, vals = oracle_csr(oracle, case)
    ok, slots, dict_entries, code_bytes = capi.host_sell_check(rp, cols)
    n_chunks = (n + 511) // 512
    assert ok and code_bytes > 512 * n_chunks          # not the 2-bytes-per-thread pattern mode
    s = reg.solver("sell_offsets", cfg(1)).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    x = rng.uniform(-1, 1, n)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


def test_irregular_pattern_is_timed_on_both_kernels_once(reg, oracle):
    """A hex mesh in a shuffled numbering, above the tuning size: after RCM the compressed layout qualifies
    with 16-bit deltas, but whether its slot-major gather beats the CSR-stream kernel's row-major one is
    measured once per pattern.  Whichever runs, the bits are the oracle's; coefficient updates keep working
    on the layout that was chosen; `force` skips the measurement."""
    from helpers import oracle_matrix_renumbered, to_new
    case = synthetic.renumber_case(synthetic.poisson_case(64), 65536)   # (some chunks need 16-bit deltas)
    rng = np.random.default_rng(11)
    case.upper[:] = rng.uniform(-1, -0.5, case.upper.size)
    x = rng.uniform(-1, 1, case.n_cells)
    s = reg.solver("sell_tuned", cfg(1)).set_matrix(case)
    LAYOUT_CSR21 = 3.0   # CSR-stream reading its columns as packed 21-bit offsets: the third candidate
    t_csr, t_sell, t_21 = (s.get_property(k) for k in ("spmvTunedCsrUs", "spmvTunedSellUs", "spmvTunedCsr21Us"))
    assert t_csr > 0 and t_sell > 0 and t_21 > 0
    best = min(t_csr, t_sell, t_21)
    assert {LAYOUT_SELL: t_sell, LAYOUT_CSR21: t_21, LAYOUT_CSR: t_csr}[s.get_property("spmvLayout")] == best
    forced = reg.solver("sell_forced", cfg(2)).set_matrix(case)
    assert forced.get_property("spmvLayout") == LAYOUT_SELL
    new_id = s.renumbering()
    assert new_id is not None and np.array_equal(new_id, forced.renumbering())
    for round_ in range(2):
        A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
        ref = oracle.spmv(rp, cols, vals, to_new(x, new_id))[new_id]
        np.testing.assert_array_equal(s.spmv(x), ref)
        np.testing.assert_array_equal(forced.spmv(x), ref)
        # next time step: new coefficients, same pattern -> no new measurement, same layout
        case.upper[:] = rng.uniform(-1, -0.5, case.upper.size)
        case.diag[:] += 0.25
        s.set_matrix(case)
        forced.set_matrix(case)
        assert s.get_property("spmvTunedCsrUs") == t_csr and s.get_property("spmvTunedSellUs") == t_sell
    # a polyhedral mesh: the chunked ELL is not even tried (plain RCM order); the CSR-stream kernel runs, on the
    # plain or on the packed columns, whichever measured faster; same bits
    vor = synthetic.voronoi_case(70000)
    sv = reg.solver("sell_voronoi", cfg(1)).set_matrix(vor)
    assert sv.get_property("spmvLayout") in (LAYOUT_CSR, LAYOUT_CSR21) and sv.get_property("rowsSortedByLength") == 0.0
    assert sv.get_property("spmvTunedCsr21Us") > 0
    nid = sv.renumbering()
    xv = rng.uniform(-1, 1, vor.n_cells)
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, vor, nid)
    np.testing.assert_array_equal(sv.spmv(xv), oracle.spmv(rp, cols, vals, to_new(xv, nid))[nid])


@pytest.mark.parametrize("append", [False, True])
def test_octree_mesh_same_bits_as_the_oracle(reg, oracle, append):
    """A hex-dominant (octree) mesh: rows of 7 entries and, along the refined shell, 10 / 13 / 16.  Lanes stop
    at their rows' ends, long tails spill; SpMV and whole CG / BiCGStab solves are the oracle's bit for bit."""
    case = synthetic.octree_case(28, 1.5, append)
    rng = np.random.default_rng(5)
    case.upper[:] = rng.uniform(-1.0, -0.5, case.upper.size)
    s = reg.solver(f"octree{int(append)}", cfg(1, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert s.get_property("spmvLayout") == LAYOUT_SELL
    rp, cols, vals = oracle_csr(oracle, case)
    x = rng.uniform(-1, 1, case.n_cells)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
    b = oracle.spmv(rp, cols, vals, x)
    A, _ = oracle_matrix(oracle, case)
    for solver, fn, kw in ((capi.SOLVER_CG, oracle.cg, {}), (capi.SOLVER_BICGSTAB, oracle.bicgstab, {})):
        sv = reg.solver(f"octree{int(append)}_{solver}", cfg(1, solver=solver, renumber=capi.RENUMBER_OFF, max_iter=40)).set_matrix(case)
        xs, perf = sv.solve(b, np.zeros_like(b))
        with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
            ref = fn(A, b, np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=1e-11,
                     rel_tol=0.0, max_iter=40, **kw)
        np.testing.assert_array_equal(sv.history(), ref.history)
        np.testing.assert_array_equal(xs, ref.x)
