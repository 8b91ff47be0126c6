"""GPU parity on the unstructured proxies of bench.py at sizes where every mechanism is in play (>= 50 k rows;
Voronoi >= 100 k cells): the library's own renumbering (RCM, rows of a wavefront sorted by length), 16-bit delta /
32-bit column codes, lane-level row ends, the spill list, the one-off timing that picks the SpMV kernel of an
irregular pattern.  Whatever layout and numbering the library ends up with, the SpMV and a 12-turn GKOCG + BJ
history must be bit-identical to the oracle run on the system permuted by the numbering the library REPORTS
(an explicit input of the oracle), in the device's reduction tree.  """
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import oracle_matrix, blocked, oracle_csr, oracle_matrix_renumbered, to_new

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


_cache = {}


def proxy(name):
    if name not in _cache:
        _cache[name] = {
            "octree": lambda: synthetic.octree_case(40, 1.5),
            "octree_append": lambda: synthetic.octree_case(40, 4.0, True),
            "long_rows": lambda: synthetic.long_rows_case(synthetic.poisson_case(40), 0.03, 40),
            "long_rows_shuffled": lambda: synthetic.renumber_case(
                synthetic.long_rows_case(synthetic.poisson_case(40), 0.03, 40), 4096),
            "drop_faces": lambda: synthetic.drop_faces_case(synthetic.poisson_case(40), 0.3),
            "drop_faces_shuffled": lambda: synthetic.renumber_case(
                synthetic.drop_faces_case(synthetic.poisson_case(40), 0.3), 4096),
            "shuffled": lambda: synthetic.renumber_case(synthetic.poisson_case(44), 65536),
            "voronoi": lambda: synthetic.voronoi_case(110000),
            # the same mesh handed over WITH its cell centres (ogl_ldu_view::cell_centres): the Hilbert order through
            # them competes with reverse Cuthill-McKee for the numbering of the device copy, and wins here
            "voronoi_centres": lambda: synthetic.voronoi_case(110000, with_centres=True),
        }[name]()
    return _cache[name]


PROXIES = ["octree", "octree_append", "long_rows", "long_rows_shuffled", "drop_faces", "drop_faces_shuffled",
           "shuffled", "voronoi", "voronoi_centres"]
# compressIndices: 1 = the default policy (irregular patterns: both kernels timed once, the faster runs),
# 2 = force the compressed layout when it qualifies, 0 = plain CSR-stream
MODES = {"auto": 1, "force": 2, "csr": 0}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", PROXIES)
def test_spmv_and_cg_history_bit_identical_to_the_oracle(reg, oracle, name, mode):
    case = proxy(name)
    assert case.n_cells >= 50000
    cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=0.0, rel_tol=0.0,
                              max_iter=12, export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0,
                              compress_indices=MODES[mode], renumber=capi.RENUMBER_AUTO)
    s = reg.solver(f"proxy_{name}_{mode}", cfg).set_matrix(case)
    new_id = s.renumbering()
    if new_id is None:
        new_id = np.arange(case.n_cells)
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    d_rp, d_cols, _, d_vals = s.local_matrix()
    np.testing.assert_array_equal(d_rp, rp)
    np.testing.assert_array_equal(d_cols, cols)
    np.testing.assert_array_equal(d_vals, vals)
    rng = np.random.default_rng(20241016)
    x = rng.uniform(-1, 1, case.n_cells)
    np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, to_new(x, new_id))[new_id])
    b = rng.uniform(-1, 1, case.n_cells)
    xs, perf = s.solve(b, x.copy())
    with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
        ref = oracle.cg(A, to_new(b, new_id), to_new(x, new_id), oracle.jacobi_generate_scalar(rp, cols, vals),
                        tolerance=0.0, rel_tol=0.0, max_iter=12)
    assert perf.n_iterations == ref.n_iterations == 13
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(xs, ref.x[new_id])
    # the same operator as the un-renumbered reference matrix, at rounding level
    o_rp, o_cols, o_vals = oracle_csr(oracle, case)
    np.testing.assert_allclose(s.spmv(x), oracle.spmv(o_rp, o_cols, o_vals, x), rtol=1e-12, atol=1e-12)


def test_the_mechanisms_are_really_in_play(reg):
    """The cases above are only worth their time if they reach the code they are meant for."""
    got = {}
    for name in PROXIES:
        case = proxy(name)
        cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, matrix_format=capi.FORMAT_CSR,
                                  compress_indices=2, renumber=capi.RENUMBER_AUTO)
        s = reg.solver(f"proxy_{name}_force", cfg).set_matrix(case)
        def prop(k):
            try:
                return s.get_property(k)
            except capi.OglError:      # (layout properties exist only once that layout was built)
                return 0.0
        got[name] = {k: prop(k) for k in ("renumbered", "rowsSortedByLength", "sellSpilledEntries",
                                          "sellChunksDelta16", "sellChunksCol32", "spmvLayout", "renumberedAlongCurve",
                                          "gatherSectorRatio")}
    assert got["voronoi"]["renumbered"] == 1.0 and got["shuffled"]["renumbered"] == 1.0, got
    assert got["voronoi"]["renumberedAlongCurve"] == 0.0 and got["voronoi_centres"]["renumberedAlongCurve"] == 1.0, got
    assert got["voronoi_centres"]["gatherSectorRatio"] < 0.8 * got["voronoi"]["gatherSectorRatio"], got
    assert got["voronoi_centres"]["rowsSortedByLength"] == 0.0, got      # (the curve's order is kept as it is)
    assert got["shuffled"]["spmvLayout"] == 2.0 and got["octree"]["spmvLayout"] == 2.0, got
    assert got["long_rows"]["sellSpilledEntries"] > 0 or got["long_rows_shuffled"]["sellSpilledEntries"] > 0, got
    assert any(g["rowsSortedByLength"] == 1.0 for g in got.values()), got
    assert any(g["sellChunksDelta16"] + g["sellChunksCol32"] > 0 for g in got.values()), got


def test_packed_columns_of_the_csr_stream_kernel(reg, oracle):
    """A polyhedral mesh does not qualify for the chunked ELL; with compressIndices the CSR-stream kernel then reads
    its columns as 21-bit offsets packed six to a word (k_spmv_stream21).  Plain, residual and two-dot (BiCGStab)
    instantiations, with and without the cache-bypassing loads, against the oracle on the reported numbering."""
    import copy
    base = proxy("voronoi")
    rng = np.random.default_rng(7)
    for asym in (False, True):
        case = copy.deepcopy(base)
        case.upper[:] = rng.uniform(-1.0, -0.25, case.upper.size)
        case.diag[:] = rng.uniform(20.0, 24.0, case.n_cells)
        if asym:
            case.lower = rng.uniform(-1.0, -0.25, case.upper.size)
        for stream in (0.0, 1e18):
            cfg = capi.default_config(solver=capi.SOLVER_BICGSTAB if asym else capi.SOLVER_CG,
                                      preconditioner=capi.PRECOND_BJ, tolerance=0.0, rel_tol=0.0, max_iter=10,
                                      export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0,
                                      compress_indices=2, renumber=capi.RENUMBER_AUTO)
            s = reg.solver(f"packed_{int(asym)}_{int(stream > 0)}", cfg)
            s.set_property("streamAboveBytes", stream)
            s.set_matrix(case)
            assert s.get_property("spmvLayout") == 3.0
            new_id = s.renumbering()
            A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
            x = rng.uniform(-1, 1, case.n_cells)
            np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, to_new(x, new_id))[new_id])
            b = rng.uniform(-1, 1, case.n_cells)
            xs, perf = s.solve(b, x.copy())
            fn = oracle.bicgstab if asym else oracle.cg
            with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
                ref = fn(A, to_new(b, new_id), to_new(x, new_id), oracle.jacobi_generate_scalar(rp, cols, vals),
                         tolerance=0.0, rel_tol=0.0, max_iter=10)
            np.testing.assert_array_equal(s.history(), ref.history)
            np.testing.assert_array_equal(xs, ref.x[new_id])


def test_packed_columns_with_far_entries_above_two_million_rows(reg, oracle):
    """Round 5: a chunk whose columns span 2^21 or more (a numbering along a space-filling curve above 2 M rows: a few
    neighbours in blobs anywhere) keeps the packed 21-bit columns -- a window of 2^21 columns around its own rows, the
    entries outside it listed apart and their products put right by the chunk's workgroup before the rows are summed.
    A chain of 2.4 M cells with 3000 couplings across more than 2^21 rows (both triangles: far entries below and above
    the window) and 2000 medium ones; plain, residual and two-dot instantiations, STREAM on and off, against the oracle."""
    n = (1 << 21) + 300000
    rng = np.random.default_rng(11)
    own = np.arange(n - 1, dtype=np.int64)
    far_lo = rng.choice(200000, 3000, replace=False).astype(np.int64)
    far_up = far_lo + (1 << 21) + rng.integers(0, 90000, far_lo.size)
    # (medium couplings: 1500 anywhere, 500 out of the first chunk's rows -- more than 255 distinct offsets in one chunk, so
    #  the compressed layout needs 32-bit columns there, the pattern counts as irregular and the packed columns are built)
    mid_lo = np.concatenate([rng.choice(n - 700000, 1500, replace=False), rng.integers(0, 512, 500)]).astype(np.int64)
    mid_up = mid_lo + rng.integers(2, 600000, mid_lo.size)
    lo = np.concatenate([own, far_lo, mid_lo])
    up = np.concatenate([own + 1, far_up, mid_up])
    assert (up < n).all()
    key = lo * n + up
    _, first = np.unique(key, return_index=True)
    lo, up = lo[first], up[first]                                  # (sorted by owner, then neighbour; duplicates dropped)
    F = lo.size
    for asym in (False, True):
        case = synthetic.LduCase(n, lo.astype(np.int32), up.astype(np.int32), rng.uniform(8.0, 9.0, n),
                                 rng.uniform(-1.0, -0.25, F), rng.uniform(-1.0, -0.25, F) if asym else None)
        for stream in (0.0, 1e18):
            cfg = capi.default_config(solver=capi.SOLVER_BICGSTAB if asym else capi.SOLVER_CG,
                                      preconditioner=capi.PRECOND_BJ, tolerance=0.0, rel_tol=0.0, max_iter=6,
                                      export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0,
                                      compress_indices=1, renumber=capi.RENUMBER_OFF,
                                      symmetric_half=0)            # (the full-storage path: what an asymmetric or
                                                                   #  irregular matrix runs on)
            s = reg.solver(f"far_{int(asym)}_{int(stream > 0)}", cfg)
            s.set_property("streamAboveBytes", stream)
            s.set_property("spmvForceLayout", 2.0)                 # the packed columns, whatever the one-off timing says
            s.set_matrix(case)
            assert s.get_property("spmvLayout") == 3.0
            assert 2 * 3000 * 0.9 < s.get_property("csr21FarEntries") <= 2 * (3000 + 2000)
            A, (rp, cols, vals) = oracle_matrix(oracle, case)
            x = rng.uniform(-1, 1, n)
            np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))
            b = rng.uniform(-1, 1, n)
            xs, perf = s.solve(b, x.copy())
            fn = oracle.bicgstab if asym else oracle.cg
            with blocked(oracle, capi.lib().ogl_reduction_chunk_rows()):
                ref = fn(A, b, x.copy(), oracle.jacobi_generate_scalar(rp, cols, vals), tolerance=0.0, rel_tol=0.0,
                         max_iter=6)
            np.testing.assert_array_equal(s.history(), ref.history)
            np.testing.assert_array_equal(xs, ref.x)
