"""GPU tests of the built-in renumbering (config `renumber`).

The product reports the permutation it chose (ogl_solver_get_renumbering); the oracle is handed the
reference-order matrix permuted by it (oracle.permute_csr -- an explicit input, not a second RCM), so
every comparison against the oracle run in the device's reduction tree stays BIT-EXACT.  Against the
un-renumbered run the results agree at rounding level only (the rows are summed in a different
column order, the reductions in a different row order) -- exactly as after OpenFOAM's renumberMesh.
"""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_csr, oracle_matrix, oracle_matrix_renumbered, oracle_precond_renumbered, to_new

pytestmark = pytest.mark.gpu

SEED = 20241016


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


@pytest.fixture(scope="module")
def chunk_rows():
    return capi.lib().ogl_reduction_chunk_rows()


def cfg(**kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_NONE, tolerance=0.0, rel_tol=0.0,
                max_iter=50, export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0,
                renumber=capi.RENUMBER_ON)
    base.update(kw)
    return capi.default_config(**base)


CASES = [
    ("box5x4x3", lambda: synthetic.poisson_block(5, 4, 3)),
    ("box_asym", lambda: synthetic.poisson_block(5, 4, 3, symmetric=False, off_upper=-0.9, off_lower=-1.1)),
    ("periodic_asym", lambda: synthetic.poisson_block(6, 5, 4, periodic_x=True, symmetric=False,
                                                      off_upper=-0.9, off_lower=-1.1)),
    ("cube33_ragged", lambda: synthetic.poisson_block(33, 31, 29)),
    ("shuffled20", lambda: synthetic.renumber_case(synthetic.poisson_case(20), 1000)),
    ("random", lambda: synthetic.random_global_case(1500, 3, 900, symmetric=False, seed=5)),
    ("line", lambda: synthetic.poisson_block(700, 1, 1)),
]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("fmt", [capi.FORMAT_CSR, capi.FORMAT_ELL], ids=["Csr", "Ell"])
def test_matrix_and_spmv(reg, oracle, name, make, fmt):
    case = make()
    s = reg.solver(f"rn_{name}_{fmt}", cfg(matrix_format=fmt)).set_matrix(case)
    new_id = s.renumbering()
    assert new_id is not None and s.get_property("renumbered") == 1.0
    assert sorted(new_id.tolist()) == list(range(case.n_cells))
    _, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    d_rp, d_cols, _, d_vals = s.local_matrix()
    np.testing.assert_array_equal(d_rp, rp)
    np.testing.assert_array_equal(d_cols, cols)
    np.testing.assert_array_equal(d_vals, vals)
    x = np.random.default_rng(SEED).uniform(-1, 1, case.n_cells)
    y = s.spmv(x)                                   # caller's order in, caller's order out
    np.testing.assert_array_equal(y, oracle.spmv(rp, cols, vals, to_new(x, new_id))[new_id])
    # and the same operator as the un-renumbered reference matrix, at rounding level
    o_rp, o_cols, o_vals = oracle_csr(oracle, case)
    np.testing.assert_allclose(y, oracle.spmv(o_rp, o_cols, o_vals, x), rtol=1e-13, atol=1e-13)


SOLVES = [
    ("cg_none", dict(solver=capi.SOLVER_CG), True),
    ("cg_bj", dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ), True),
    ("cg_bj4", dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, max_block_size=4), True),
    ("cg_isai", dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_ISAI), True),
    ("bicg_bj", dict(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_BJ), False),
    ("bicg_gisai", dict(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_GISAI), False),
    ("gmres_bj", dict(solver=capi.SOLVER_GMRES, preconditioner=capi.PRECOND_BJ, krylov_dim=10), False),
]


@pytest.mark.parametrize("name,kw,sym", SOLVES, ids=[c[0] for c in SOLVES])
def test_solver_history_bit_identical_to_oracle_on_the_permuted_system(reg, oracle, chunk_rows, name, kw, sym):
    case = synthetic.renumber_case(synthetic.poisson_case(14, symmetric=sym), 700)
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    skw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=300)
    s = reg.solver("rns_" + name, cfg(**kw, **skw)).set_matrix(case)
    new_id = s.renumbering()
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    pk = kw.get("preconditioner", capi.PRECOND_NONE)
    if pk == capi.PRECOND_NONE:
        P = None
    elif pk == capi.PRECOND_BJ:
        P = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, kw.get("max_block_size", 1))
    else:
        P = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id,
                                      isai="spd" if pk == capi.PRECOND_ISAI else "general")
    fn = {capi.SOLVER_CG: oracle.cg, capi.SOLVER_BICGSTAB: oracle.bicgstab}.get(kw["solver"])
    with blocked(oracle, chunk_rows):
        if fn:
            ref = fn(A, to_new(b, new_id), np.zeros_like(b), P, **skw)
        else:
            ref = oracle.gmres(A, to_new(b, new_id), np.zeros_like(b), P, krylov_dim=10, **skw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x[new_id])
    assert perf.norm_factor == ref.norm_factor
    np.testing.assert_allclose(x, xs, atol=1e-8, rtol=0)
    # against the un-renumbered run: same answer, history equal at rounding level at the start
    s0 = reg.solver("rns0_" + name, cfg(**kw, **skw, renumber=capi.RENUMBER_OFF)).set_matrix(case)
    x0, perf0 = s0.solve(b, np.zeros_like(b))
    # the SAME operator and the SAME preconditioner in both numberings (block-Jacobi blocks and ISAI(spd)'s triangle
    # are those of the caller's numbering either way): histories equal at rounding level, iteration counts equal
    np.testing.assert_allclose(s.history()[:5], s0.history()[:5], rtol=1e-10)
    assert abs(perf.n_iterations - perf0.n_iterations) <= 1
    np.testing.assert_allclose(x, x0, atol=1e-8, rtol=0)


def test_auto_renumbers_a_badly_numbered_mesh_only(reg, oracle, chunk_rows):
    box = synthetic.poisson_case(28)                              # 21,952 rows
    s = reg.solver("rn_auto_box", cfg(renumber=capi.RENUMBER_AUTO)).set_matrix(box)
    assert s.renumbering() is None and s.get_property("spmvLayout") == 2.0
    sh = synthetic.renumber_case(box, 4096)
    s = reg.solver("rn_auto_sh", cfg(renumber=capi.RENUMBER_AUTO, preconditioner=capi.PRECOND_BJ,
                                     max_iter=40)).set_matrix(sh)
    new_id = s.renumbering()
    assert new_id is not None
    assert s.get_property("gatherSectorRatioNatural") > 0.5 > 0.25 > s.get_property("gatherSectorRatio")
    b = synthetic.apply_case(sh, synthetic.x_star(sh.global_index, sh.global_n))
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, sh, new_id)
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals),
                        tolerance=0.0, rel_tol=0.0, max_iter=40)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x[new_id])


def test_changing_the_keyword_rebuilds_the_pattern(reg, oracle):
    case = synthetic.renumber_case(synthetic.poisson_case(10), 200)
    s = reg.solver("rn_toggle", cfg(renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert s.renumbering() is None
    np.testing.assert_array_equal(s.local_matrix()[1], oracle_csr(oracle, case)[1])
    s = reg.solver("rn_toggle", cfg(renumber=capi.RENUMBER_ON)).set_matrix(case)
    assert s.renumbering() is not None
    s = reg.solver("rn_toggle", cfg(renumber=capi.RENUMBER_OFF)).set_matrix(case)
    assert s.renumbering() is None
    np.testing.assert_array_equal(s.local_matrix()[1], oracle_csr(oracle, case)[1])


def test_vectors_cross_the_boundary_in_the_callers_order(reg, oracle, chunk_rows):
    """updateInitGuess false: the previous DEVICE solution is the next initial guess
    (lduLduBase.H:235); upload / download / resident entry points permute on the way."""
    case = synthetic.renumber_case(synthetic.poisson_case(9), 100)
    rng = np.random.default_rng(SEED)
    b1, b2 = rng.uniform(-1, 1, case.n_cells), rng.uniform(-1, 1, case.n_cells)
    c = cfg(max_iter=12)
    s = reg.solver("rn_vec", c).set_matrix(case)
    new_id = s.renumbering()
    x1, _ = s.solve(b1, np.zeros_like(b1))
    x2, _ = reg.solver("rn_vec", c).set_matrix(case).solve(b2, np.full_like(b2, 9.0))  # psi ignored
    A, _ = oracle_matrix_renumbered(oracle, case, new_id)
    with blocked(oracle, chunk_rows):
        r1 = oracle.cg(A, to_new(b1, new_id), np.zeros_like(b1), None, tolerance=0.0, rel_tol=0.0, max_iter=12)
        r2 = oracle.cg(A, to_new(b2, new_id), r1.x, None, tolerance=0.0, rel_tol=0.0, max_iter=12)
    np.testing.assert_array_equal(x1, r1.x[new_id])
    np.testing.assert_array_equal(x2, r2.x[new_id])
    np.testing.assert_array_equal(s.download_solution(), x2)
    s.upload_solution(x1)
    np.testing.assert_array_equal(s.download_solution(), x1)
    s.upload_rhs(b1)
    s.upload_solution(None)
    s.apply_resident()
    np.testing.assert_array_equal(s.download_solution(), x1)


def test_export_is_written_in_the_callers_numbering(reg, tmp_path):
    case = synthetic.renumber_case(synthetic.poisson_block(7, 6, 5, periodic_x=True, symmetric=False,
                                                           off_upper=-0.9, off_lower=-1.1), 50)
    b = np.arange(case.n_cells, dtype=np.float64)
    files = {}
    for mode in (capi.RENUMBER_OFF, capi.RENUMBER_ON):
        s = reg.solver(f"rn_exp{mode}", cfg(renumber=mode, max_iter=2)).set_matrix(case)
        s.solve(b, np.zeros_like(b))
        d = str(tmp_path / f"m{mode}")
        s.export_system(d)
        files[mode] = {n: open(f"{d}/rn_exp{mode}_{n}.mtx").read() for n in ("A_local", "rhs_b_")}
    assert files[0]["A_local"] == files[1]["A_local"]
    assert files[0]["rhs_b_"] == files[1]["rhs_b_"]


def test_mixed_row_lengths_run_on_the_compressed_layout_after_the_length_sort(reg, oracle, chunk_rows):
    """Rows of 1..7 entries (a hex mesh that lost 30 % of its faces; a stand-in for mixed cell types): in the
    caller's order the 128-byte lines of the planes mix long and short rows and the padding disqualifies the
    compressed layout (CSR-stream kernel); with `renumber` the rows of every wavefront go longest first, the
    lanes stop loading at the end of their own rows, and everything stays bit-identical to the oracle on the
    permuted system."""
    case = synthetic.drop_faces_case(synthetic.poisson_case(28), 0.3)
    rng = np.random.default_rng(SEED)
    x = rng.uniform(-1, 1, case.n_cells)
    s0 = reg.solver("rn_mixed_off", cfg(renumber=capi.RENUMBER_OFF)).set_matrix(case)
    # (since round 3 a symmetric matrix in the caller's order is first tried on the half storage with per-chunk
    #  distances, which this mesh -- the box's distances with holes -- qualifies for; full storage: CSR-stream)
    assert s0.get_property("spmvLayout") == 2.0 and s0.get_property("symmetricHalfPerChunk") == 1.0
    s0f = reg.solver("rn_mixed_off_full", cfg(renumber=capi.RENUMBER_OFF, symmetric_half=0)).set_matrix(case)
    assert s0f.get_property("spmvLayout") == 0.0
    for mode in (capi.RENUMBER_ON, capi.RENUMBER_AUTO):
        s = reg.solver(f"rn_mixed_{mode}", cfg(renumber=mode, preconditioner=capi.PRECOND_BJ, max_iter=30)).set_matrix(case)
        new_id = s.renumbering()
        assert new_id is not None and s.get_property("spmvLayout") == 2.0
        assert s.get_property("rowsSortedByLength") == 1.0
        assert s.get_property("sellReadSlots") < 1.15 * (case.nnz + 4096) < s.get_property("sellAllocatedSlots")
        A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
        lens = np.diff(rp)
        assert all(np.all(np.diff(lens[c:c + 128]) <= 0) for c in range(0, case.n_cells, 128))
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, to_new(x, new_id))[new_id])
        np.testing.assert_allclose(s.spmv(x), s0.spmv(x), rtol=1e-13, atol=1e-13)
        b = synthetic.apply_case(case, x)
        xs, perf = s.solve(b, np.zeros_like(b))
        with blocked(oracle, chunk_rows):
            ref = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), oracle.jacobi_generate_scalar(rp, cols, vals),
                            tolerance=0.0, rel_tol=0.0, max_iter=30)
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(xs, ref.x[new_id])


@pytest.mark.parametrize("name,kw", [("cg_bj4", dict(preconditioner=capi.PRECOND_BJ, max_block_size=4)),
                                     ("cg_isai", dict(preconditioner=capi.PRECOND_ISAI))])
def test_auto_renumbering_above_its_size_threshold_with_numbering_dependent_preconditioners(reg, oracle, chunk_rows,
                                                                                            name, kw):
    """renumber auto at >= 16384 rows: block-Jacobi blocks (runs of consecutive rows)
    and ISAI(spd)'s tril(A) are those of the CALLER's numbering -- the matrix OpenFOAM hands over is what the
    reference generates its preconditioner on (Preconditioner.H:91-105, :225-241) -- carried through the
    permutation, so the backend's renumbering changes neither the operator nor the preconditioner.  Bit-exact
    against the oracle on the permuted system with that preconditioner; against `renumber off` the same iteration
    count (+-1) and the same leading history at rounding level."""
    case = synthetic.renumber_case(synthetic.poisson_case(28), 4096)          # 21,952 rows
    xs = synthetic.x_star(case.global_index, case.global_n)
    b = synthetic.apply_case(case, xs)
    skw = dict(tolerance=1e-10, rel_tol=0.0, max_iter=400)
    s = reg.solver("rna_" + name, cfg(solver=capi.SOLVER_CG, renumber=capi.RENUMBER_AUTO, **kw, **skw)).set_matrix(case)
    new_id = s.renumbering()
    assert new_id is not None and s.get_property("renumbered") == 1.0
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    P = (oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, kw["max_block_size"])
         if kw["preconditioner"] == capi.PRECOND_BJ
         else oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, isai="spd"))
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), P, **skw)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x[new_id])
    s0 = reg.solver("rna0_" + name, cfg(solver=capi.SOLVER_CG, renumber=capi.RENUMBER_OFF, **kw, **skw)).set_matrix(case)
    x0, perf0 = s0.solve(b, np.zeros_like(b))
    np.testing.assert_allclose(x, x0, atol=1e-8, rtol=0)
    np.testing.assert_allclose(x, xs, atol=1e-8, rtol=0)
    print(f"{name}: {perf.n_iterations} iterations in the backend's numbering, {perf0.n_iterations} in the mesh's")
    assert abs(perf.n_iterations - perf0.n_iterations) <= 1
    np.testing.assert_allclose(s.history()[:20], s0.history()[:20], rtol=1e-9)
    # ... and the same count as the oracle on the CALLER's numbering (sequential order: the reference executor)
    A0, (rp0, cols0, vals0) = oracle_matrix(oracle, case)
    P0 = (oracle.Precond(rp0, cols0, vals0, kw["max_block_size"]) if kw["preconditioner"] == capi.PRECOND_BJ
          else oracle.Precond(rp0, cols0, vals0, isai="spd"))
    ref0 = oracle.cg(A0, b, np.zeros_like(b), P0, **skw)
    assert abs(perf.n_iterations - ref0.n_iterations) <= 1


@pytest.mark.parametrize("name,kw", [("cg_bj4", dict(preconditioner=capi.PRECOND_BJ, max_block_size=4)),
                                     ("cg_isai", dict(preconditioner=capi.PRECOND_ISAI))])
def test_preconditioner_structures_in_the_backends_numbering_on_request(reg, oracle, chunk_rows, name, kw):
    """Property precondCallerNumbering 0 (the A/B switch): blocks / triangle taken in the backend's
    numbering -- the round-3 behaviour, another preconditioner of the same kind; bit-exact against the oracle that
    builds its preconditioner on the permuted system.  Flipping the property on a live field rebuilds the structure."""
    case = synthetic.renumber_case(synthetic.poisson_case(14), 700)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    skw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=300)
    s = reg.solver("rnb_" + name, cfg(**kw, **skw))
    s.set_property("precondCallerNumbering", 0.0)
    s.set_matrix(case)
    new_id = s.renumbering()
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    P_dev = (oracle.Precond(rp, cols, vals, kw["max_block_size"]) if kw["preconditioner"] == capi.PRECOND_BJ
             else oracle.Precond(rp, cols, vals, isai="spd"))
    P_caller = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, kw.get("max_block_size", 1),
                                         isai="spd" if kw["preconditioner"] == capi.PRECOND_ISAI else None)
    with blocked(oracle, chunk_rows):
        ref_dev = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), P_dev, **skw)
        ref_caller = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), P_caller, **skw)
    np.testing.assert_array_equal(s.history(), ref_dev.history)
    np.testing.assert_array_equal(x, ref_dev.x[new_id])
    assert not np.array_equal(ref_dev.history[:10], ref_caller.history[:10])      # really two operators
    s.set_property("precondCallerNumbering", 1.0)                                 # back to the reference's operator
    s.upload_solution(None)
    x1, _ = s.solve(b, np.zeros_like(b))
    np.testing.assert_array_equal(s.history(), ref_caller.history)
    np.testing.assert_array_equal(x1, ref_caller.x[new_id])


@pytest.mark.parametrize("staged", [1.0, 2.0, 0.0], ids=["one_pass", "staged", "direct"])
@pytest.mark.parametrize("k", [2, 4, 7, 32])
def test_block_jacobi_through_the_permutation_both_applies(reg, oracle, chunk_rows, staged, k):
    """The caller's blocks on a renumbered copy: the one-pass apply over the blocks in the caller's order (every
    workgroup gathers its positions' inputs once into LDS, blocks that straddle its range included; default), the
    staged one (vectors carried into the caller's order and back by kernels of their own, bjFusedPerm 0) and the
    direct one (block rows stored at their device rows, members gathered) give the oracle's bits; block sizes that do
    not divide the row count, up to the largest (32: the widest halo of the one-pass form)."""
    case = synthetic.renumber_case(synthetic.poisson_block(13, 11, 7 if k < 32 else 19), 300)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    skw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=200)
    s = reg.solver(f"rnbj_{k}_{staged}", cfg(preconditioner=capi.PRECOND_BJ, max_block_size=k, **skw))
    s.set_property("bjStagedApply", 1.0 if staged else 0.0)
    s.set_property("bjFusedPerm", 1.0 if staged == 1.0 else 0.0)
    s.set_matrix(case)
    new_id = s.renumbering()
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    P = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, k)
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), P, **skw)
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x[new_id])
    # BiCGStab and GMRES use the apply without the fused dot
    for solver, fn, extra in ((capi.SOLVER_BICGSTAB, oracle.bicgstab, {}), (capi.SOLVER_GMRES, oracle.gmres, dict(krylov_dim=10))):
        s2 = reg.solver(f"rnbj_{k}_{staged}_{solver}", cfg(solver=solver, preconditioner=capi.PRECOND_BJ, max_block_size=k,
                                                          krylov_dim=extra.get("krylov_dim", 0), **skw))
        s2.set_property("bjStagedApply", 1.0 if staged else 0.0)
        s2.set_property("bjFusedPerm", 1.0 if staged == 1.0 else 0.0)
        s2.set_matrix(case)
        x2, _ = s2.solve(b, np.zeros_like(b))
        with blocked(oracle, chunk_rows):
            ref2 = fn(A, to_new(b, new_id), np.zeros_like(b), P, **extra, **skw)
        np.testing.assert_array_equal(s2.history(), ref2.history)
        np.testing.assert_array_equal(x2, ref2.x[new_id])


@pytest.mark.parametrize("name,kw", [("bj1", dict(preconditioner=capi.PRECOND_BJ)),
                                     ("gisai", dict(preconditioner=capi.PRECOND_GISAI)),
                                     ("bj3_backend", dict(preconditioner=capi.PRECOND_BJ, max_block_size=3))])
def test_stored_preconditioner_of_another_numbering_is_not_applied(oracle, chunk_rows, name, kw):
    """the preconditioner store is shared by all fields of a registry (Preconditioner.H:357).  A stored
    inverse diagonal, W, or block Jacobi whose VALUES are laid out in the device numbering of the field that generated
    it would be a silently permuted operator for a field whose device copy is numbered differently: such an object is
    regenerated for the solve instead (`PrecondData::foreign_to`), and the second field's history is the oracle's with
    ITS OWN preconditioner.  (Blocks kept block-major in the caller's order stay portable, as in the reference: the
    applying solver carries the vectors through its permutation.)"""
    a = synthetic.renumber_case(synthetic.poisson_case(14), 700, seed=1)
    b_case = synthetic.renumber_case(synthetic.poisson_case(14), 700, seed=2)            # same sizes, another numbering
    skw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=200)
    r = capi.Registry()
    try:
        sols = []
        for tag, case in (("A", a), ("B", b_case)):
            s = r.solver(f"store_{name}_{tag}", cfg(caching=3, **kw, **skw))
            if name == "bj3_backend":
                s.set_property("precondCallerNumbering", 0.0)
            s.set_matrix(case)
            if tag == "B":      # (this field's counter says "use the stored object": Preconditioner.H:384-418)
                s.set_property("preconditionerCaching", 2.0)
            rhs = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
            x, perf = s.solve(rhs, np.zeros_like(rhs))
            sols.append((s, case, rhs, x))
        s, case, rhs, x = sols[1]
        new_id = s.renumbering()
        assert not np.array_equal(new_id, sols[0][0].renumbering())
        A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
        if name == "bj3_backend":
            P = oracle.Precond(rp, cols, vals, 3)                                         # the backend's own blocks
        elif kw["preconditioner"] == capi.PRECOND_BJ:
            P = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, kw.get("max_block_size", 1))
        else:
            P = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, isai="general")
        with blocked(oracle, chunk_rows):
            ref = oracle.cg(A, to_new(rhs, new_id), np.zeros_like(rhs), P, **skw)
        np.testing.assert_array_equal(s.history(), ref.history)
        np.testing.assert_array_equal(x, ref.x[new_id])
    finally:
        r.close()


@pytest.mark.parametrize("group", [0.0, 1.0, 3.0, 7.0, 1000.0])
def test_block_jacobi_through_the_permutation_in_xcd_slabs(reg, oracle, chunk_rows, group):
    """k_bj_apply_perm deals its 512-position ranges to the XCDs in slabs (bjPermXcdGroup; 0: chosen from the size): whatever
    the slab length -- one, odd, longer than the system -- every range is taken exactly once, and the bits are the oracle's
    (125 ranges: the last slab of every setting is only partly filled)."""
    case = synthetic.renumber_case(synthetic.poisson_case(40), 4096)
    b = synthetic.apply_case(case, synthetic.x_star(case.global_index, case.global_n))
    skw = dict(tolerance=1e-10, rel_tol=0.0, max_iter=60)
    s = reg.solver(f"rnbj_slab_{group}", cfg(preconditioner=capi.PRECOND_BJ, max_block_size=4, **skw))
    s.set_property("bjPermXcdGroup", group)
    s.set_matrix(case)
    new_id = s.renumbering()
    x, perf = s.solve(b, np.zeros_like(b))
    A, (rp, cols, vals) = oracle_matrix_renumbered(oracle, case, new_id)
    P = oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, 4)
    with blocked(oracle, chunk_rows):
        ref = oracle.cg(A, to_new(b, new_id), np.zeros_like(b), P, **skw)
    assert perf.n_iterations == ref.n_iterations
    np.testing.assert_array_equal(s.history(), ref.history)
    np.testing.assert_array_equal(x, ref.x[new_id])
