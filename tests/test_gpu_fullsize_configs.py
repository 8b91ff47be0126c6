"""The other BASELINE.json configs at their per-GPU sizes on one MI355X, through size-independent
properties (VERDICT r1 "configs 3, 4, 5 are exercised only at oracle-sized toys"):

  configs[2]  ~2 M cells, unstructured: GKOBiCGStab + ISAI on a momentum-like (non-symmetric) matrix,
              GKOCG + BJ on a pressure-like one            -> 128^3 box, cells renumbered at random
              inside windows of 65536 (the backend renumbers its device copy by itself)
              ... and a three-block blockMesh (pitzDaily-like: boxes of different length glued along x, each
              numbered on its own) of 2.06 M cells with the same keyword pairs
  configs[3]  20 M cells / 8 GPUs, GKOCG + BJ              -> one rank's 136^3 share
  configs[4]  50 M cells / 8 GPUs, GKOGMRES(30) + BJ, Csr vs Ell -> one rank's 184^3 share, shuffled

Properties: A.1 = delta (row sums), the product of the renumbered device copy equals the product of the
un-renumbered one at rounding level, round trip A x = A x* -> x*, the reported normalised-L1 residual
equals the one recomputed from the returned x, Csr and Ell give the same history bit for bit, the same
solve twice gives the same bits.
"""
import numpy as np
import pytest

from ogl_amd import capi, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def cfg(**kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-9, rel_tol=0.0,
                max_iter=3000, export_res=1, adapt_min_iter=0, matrix_format=capi.FORMAT_CSR)
    base.update(kw)
    return capi.default_config(**base)


def check_round_trip(s, case, tol_x=1e-6, stale_residual=False):
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = s.spmv(xs)
    s.upload_solution(None)
    x, perf = s.solve(b, np.zeros_like(b))
    hist = s.history()
    assert perf.final_residual < 1e-9
    assert hist[0] == perf.initial_residual and hist[-1] == perf.final_residual
    np.testing.assert_allclose(x, xs, rtol=0, atol=tol_x)
    r = b - s.spmv(x)
    true_res = np.abs(r).sum() / perf.norm_factor
    if stale_residual:
        # GKOGMRES hands the criterion the residual of the last restart: the returned x
        # has taken the Arnoldi steps since then, so it is at least as good as reported
        assert 0.5 * perf.final_residual < true_res <= perf.final_residual * (1 + 2e-3)
    else:
        assert true_res == pytest.approx(perf.final_residual, rel=2e-3)
    return x, perf, hist, b


def test_config2_unstructured_2m_cells(reg):
    box = synthetic.poisson_case(128)
    case = synthetic.renumber_case(box, 65536)
    s = reg.solver("p", cfg()).set_matrix(case)
    assert s.renumbering() is not None
    # an irregular pattern of this size: the three candidates (CSR-stream, chunked ELL, CSR-stream with packed
    # columns) were timed once, the fastest one runs
    t = {0.0: s.get_property("spmvTunedCsrUs"), 2.0: s.get_property("spmvTunedSellUs"),
         3.0: s.get_property("spmvTunedCsr21Us")}
    assert t[s.get_property("spmvLayout")] == min(t.values())
    assert s.get_property("gatherSectorRatioNatural") > 0.5 > 0.2 > s.get_property("gatherSectorRatio")
    delta = 1e-3 * (1.0 + (case.global_index % 7) / 7.0)
    np.testing.assert_allclose(s.spmv(np.ones(case.n_cells)), delta, rtol=0, atol=4e-15)
    # same operator as the device copy kept in the caller's numbering
    s0 = reg.solver("p_natural", cfg(renumber=capi.RENUMBER_OFF)).set_matrix(case)
    x = np.random.default_rng(3).uniform(-1, 1, case.n_cells)
    np.testing.assert_allclose(s.spmv(x), s0.spmv(x), rtol=0, atol=2e-14)
    xa, perf, hist, b = check_round_trip(s, case)
    # deterministic: the same solve again gives the same bits
    s.upload_solution(None)
    xb, perf_b = s.solve(b, np.zeros_like(b))
    np.testing.assert_array_equal(xa, xb)
    assert perf_b.n_iterations == perf.n_iterations
    # momentum-like matrix: GKOBiCGStab + ISAI (the keyword pair of configs[2]) and + GISAI
    asym = synthetic.renumber_case(synthetic.poisson_case(128, symmetric=False), 65536)
    for name, pc in (("U_isai", capi.PRECOND_ISAI), ("U_gisai", capi.PRECOND_GISAI)):
        su = reg.solver(name, cfg(solver=capi.SOLVER_BICGSTAB, preconditioner=pc)).set_matrix(asym)
        assert su.renumbering() is not None
        _, perf_u, _, _ = check_round_trip(su, asym)
    # ... and needs fewer turns than with scalar Jacobi
    sj = reg.solver("U_bj", cfg(solver=capi.SOLVER_BICGSTAB)).set_matrix(asym)
    _, perf_j, _, _ = check_round_trip(sj, asym)
    assert perf_u.n_iterations < perf_j.n_iterations


def test_config2_multi_block_2m_cells(reg):
    """configs[2] on the kind of mesh pitzDaily is: blockMesh blocks of different sizes, each numbered x-fastest
    by itself, so the distances to the y- and z-neighbours change from block to block and the faces between
    blocks couple cells at distances no band holds."""
    import dataclasses
    case = synthetic.multi_block_case([60, 90, 40], 104, 104)
    assert case.n_cells == 2055040
    s = reg.solver("p_blocks", cfg()).set_matrix(case)
    assert s.renumbering() is None               # already banded block by block: left as the caller numbered it
    # no band holds the block-to-block faces, so the whole-matrix half storage does not qualify; the per-chunk
    # one does and is timed against the compressed ELL rows -- whichever measured faster is what runs
    t_symx, t_sell = s.get_property("spmvTunedSymxUs"), s.get_property("spmvTunedSellUs")
    assert t_symx > 0 and t_sell > 0
    kept = 1.0 if t_symx <= t_sell else 0.0
    assert s.get_property("symmetricHalfPerChunk") == kept and s.get_property("symmetricHalf") == kept
    delta = 1e-3 * (1.0 + (case.global_index % 7) / 7.0)
    np.testing.assert_allclose(s.spmv(np.ones(case.n_cells)), delta, rtol=0, atol=4e-15)
    # the product does not depend on the layout the tuner picked
    s0 = reg.solver("p_blocks_full", cfg(symmetric_half=0, compress_indices=0)).set_matrix(case)
    x = np.random.default_rng(5).uniform(-1, 1, case.n_cells)
    np.testing.assert_allclose(s.spmv(x), s0.spmv(x), rtol=0, atol=2e-14)
    xa, perf, hist, b = check_round_trip(s, case)
    s.upload_solution(None)
    xb, perf_b = s.solve(b, np.zeros_like(b))
    np.testing.assert_array_equal(xa, xb)
    assert perf_b.n_iterations == perf.n_iterations
    # momentum-like matrix on the same addressing
    asym = dataclasses.replace(case, upper=np.full(case.n_faces, -0.9), lower=np.full(case.n_faces, -1.1))
    su = reg.solver("U_blocks_isai", cfg(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_ISAI)).set_matrix(asym)
    _, perf_u, _, _ = check_round_trip(su, asym)
    sj = reg.solver("U_blocks_bj", cfg(solver=capi.SOLVER_BICGSTAB)).set_matrix(asym)
    _, perf_j, _, _ = check_round_trip(sj, asym)
    assert perf_u.n_iterations < perf_j.n_iterations


def test_config3_one_rank_share_of_20m_cells(reg):
    case = synthetic.poisson_case(136)                         # 2,515,456 rows
    s = reg.solver("c3", cfg()).set_matrix(case)
    assert s.renumbering() is None and s.get_property("spmvLayout") == 2.0
    check_round_trip(s, case)


def test_config4_gmres_csr_vs_ell_one_rank_share_of_50m_cells(reg):
    case = synthetic.renumber_case(synthetic.poisson_case(184), 65536)     # 6,229,504 rows
    kw = dict(solver=capi.SOLVER_GMRES, krylov_dim=30, max_iter=6000)
    s_csr = reg.solver("c4_csr", cfg(**kw)).set_matrix(case)
    x_csr, perf_csr, hist_csr, b = check_round_trip(s_csr, case, tol_x=1e-5, stale_residual=True)
    s_ell = reg.solver("c4_ell", cfg(matrix_format=capi.FORMAT_ELL, **kw)).set_matrix(case)
    assert s_ell.get_property("spmvLayout") == 1.0 and s_csr.get_property("spmvLayout") == 2.0
    s_ell.upload_solution(None)
    x_ell, perf_ell = s_ell.solve(b, np.zeros_like(b))
    # both numberings are the plain RCM one (no length sort on a hex mesh), rows are summed in the same
    # stored order by both kernels: same bits
    np.testing.assert_array_equal(s_ell.renumbering(), s_csr.renumbering())
    np.testing.assert_array_equal(s_ell.history(), hist_csr)
    np.testing.assert_array_equal(x_ell, x_csr)
