"""CPU tests of the renumbering host logic (config `renumber`): reverse Cuthill-McKee order, the
permuted pattern against the oracle's P A P^T of the reference-order matrix, the auto policy.

The oracle takes the permutation as an explicit INPUT (oracle.permute_csr): what is checked is that
the product's renumbered pattern + ldu_mapping describe exactly the permuted reference matrix.
"""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import oracle_csr, orc_ifaces

SEED = 20241016


def rowptr_of(rows, n):
    return np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)


def source_array(case):
    """[upper | lower (asym) | diag | local-interface coefficients * -1] (HostMatrix.C:644-682)."""
    parts = [case.upper]
    if case.lower is not None:
        parts.append(case.lower)
    parts.append(case.diag)
    for f in case.interfaces:
        if f.kind == synthetic.IFACE_CYCLIC:
            parts.append(-1.0 * f.bou_coeffs)
    return np.concatenate(parts)


CASES = [
    ("box7x6x5", lambda: synthetic.poisson_block(7, 6, 5)),
    ("box7x6x5_asym", lambda: synthetic.poisson_block(7, 6, 5, symmetric=False, off_upper=-0.9, off_lower=-1.1)),
    ("periodic", lambda: synthetic.poisson_block(6, 5, 4, periodic_x=True)),
    ("periodic_asym", lambda: synthetic.poisson_block(6, 5, 4, periodic_x=True, symmetric=False,
                                                      off_upper=-0.9, off_lower=-1.1)),
    ("shuffled", lambda: synthetic.renumber_case(synthetic.poisson_case(12), 300)),
    ("random", lambda: synthetic.random_global_case(700, 3, 60, symmetric=False, seed=3)),
    ("rank1of4", lambda: synthetic.poisson_block(8, 8, 8, 2, 2, 1, 1)),
    ("single_cell", lambda: synthetic.poisson_block(1, 1, 1)),
    ("two_components", lambda: synthetic.LduCase(6, np.array([0, 1, 3, 4], np.int32),
                                                 np.array([1, 2, 4, 5], np.int32), np.full(6, 3.0),
                                                 np.full(4, -1.0), None)),
]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_renumbered_pattern_is_the_permuted_reference_matrix(oracle, name, make):
    case = make()
    n = case.n_cells
    d, loc, nl, comm, (ren, new_id) = capi.host_pattern_renumbered(case, capi.RENUMBER_ON)
    assert ren == (n >= 2)
    assert sorted(new_id.tolist()) == list(range(n))            # a permutation
    o_rp, o_cols, o_vals = oracle_csr(oracle, case)             # reference order
    p_rp, p_cols, p_vals, _ = oracle.permute_csr(o_rp, o_cols, o_vals, new_id)
    rows, cols, mapping = loc
    np.testing.assert_array_equal(rowptr_of(rows, n), p_rp)
    np.testing.assert_array_equal(cols, p_cols)
    # ldu_mapping still addresses the caller's coefficient arrays
    np.testing.assert_array_equal(source_array(case)[mapping], p_vals)
    # rows are sorted, columns ascending inside a row
    assert np.all(np.diff(rows) >= 0)
    for r in range(n):
        assert np.all(np.diff(cols[p_rp[r]:p_rp[r + 1]]) >= 0)
    # halo part
    ifs = orc_ifaces(oracle, case)
    o_nl_rows, o_nl_cols, o_nl_perm = oracle.init_non_local_sparsity(ifs)
    o_nl_vals = oracle.update_non_local_matrix_data(ifs, o_nl_perm)
    e_rows, e_cols, e_vals, order = oracle.permute_non_local(o_nl_rows, o_nl_cols, o_nl_vals, new_id)
    np.testing.assert_array_equal(nl[0], e_rows)
    np.testing.assert_array_equal(nl[1], e_cols)
    np.testing.assert_array_equal(nl[2], o_nl_perm[order])
    ids, sizes, send = oracle.create_communication_pattern(ifs)
    np.testing.assert_array_equal(comm[0], ids)
    np.testing.assert_array_equal(comm[1], sizes)
    np.testing.assert_array_equal(comm[2], new_id[send])         # same order, new names


def test_mode_off_is_the_reference_pattern(oracle):
    case = synthetic.renumber_case(synthetic.poisson_case(10), 100)
    d, loc, nl, comm, (ren, new_id) = capi.host_pattern_renumbered(case, capi.RENUMBER_OFF)
    d0, loc0, nl0, comm0 = capi.host_pattern(case)
    assert not ren
    np.testing.assert_array_equal(new_id, np.arange(case.n_cells))
    for a, b in zip(loc, loc0):
        np.testing.assert_array_equal(a, b)


def bandwidth(rows, cols):
    return int(np.abs(rows.astype(np.int64) - cols).max())


def test_rcm_cuts_bandwidth_and_sector_ratio():
    case = synthetic.renumber_case(synthetic.poisson_case(24), 4096)
    d, loc, _, _ = capi.host_pattern(case)
    rows, cols, _ = loc
    rp = rowptr_of(rows, d.n_rows)
    new_id = capi.host_rcm(rp, cols)
    assert sorted(new_id.tolist()) == list(range(d.n_rows))
    np.testing.assert_array_equal(new_id, capi.host_rcm(rp, cols))   # deterministic
    assert bandwidth(new_id[rows], new_id[cols]) < 0.5 * bandwidth(rows, cols)
    before = capi.host_gather_sector_ratio(rp, cols)
    after = capi.host_gather_sector_ratio(rp, cols, new_id)
    assert before > 0.5 and after < 0.2, (before, after)
    # the structured numbering is already good
    c0 = synthetic.poisson_case(24)
    d0, loc0, _, _ = capi.host_pattern(c0)
    assert capi.host_gather_sector_ratio(rowptr_of(loc0[0], d0.n_rows), loc0[1]) < 0.2


def test_auto_policy():
    # structured box: the compressed layout qualifies -> caller's numbering kept
    big = synthetic.poisson_case(28)                      # 21,952 rows >= RENUMBER_AUTO_MIN_ROWS
    assert not capi.host_pattern_renumbered(big, capi.RENUMBER_AUTO)[4][0]
    # the same box with shuffled numbering: renumbered
    sh = synthetic.renumber_case(big, 4096)
    d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(sh, capi.RENUMBER_AUTO)
    assert ren
    rp = rowptr_of(loc[0], d.n_rows)
    assert capi.host_gather_sector_ratio(rp, loc[1]) < 0.2
    # small systems are left alone whatever their numbering
    small = synthetic.renumber_case(synthetic.poisson_case(12), 512)
    assert not capi.host_pattern_renumbered(small, capi.RENUMBER_AUTO)[4][0]
    # without the compressed layout in play (matrixFormat Ell / compressIndices 0) the sector ratio decides
    assert not capi.host_pattern_renumbered(big, capi.RENUMBER_AUTO, compress_indices=0)[4][0]
    assert capi.host_pattern_renumbered(sh, capi.RENUMBER_AUTO, compress_indices=0)[4][0]


def test_bad_mode_is_refused():
    with pytest.raises(capi.OglError):
        capi.host_pattern_renumbered(synthetic.poisson_case(4), 7)


def test_fingerprint_sees_every_face():
    """a change of the addressing that keeps the counts must change the hash wherever it
    happens (the round-1 fingerprint sampled ~1024 entries per array)."""
    case = synthetic.poisson_block(120, 120, 100)           # 4.3 M faces: threaded path
    h0 = capi.host_addressing_fingerprint(case)
    assert h0 == capi.host_addressing_fingerprint(synthetic.poisson_block(120, 120, 100))
    rng = np.random.default_rng(SEED)
    for f in [0, case.n_faces - 1, *rng.integers(0, case.n_faces, 20).tolist()]:
        for name in ("upper_addr", "lower_addr"):
            arr = getattr(case, name)
            old = int(arr[f])
            arr[f] = old ^ 1
            assert capi.host_addressing_fingerprint(case) != h0, (name, f)
            arr[f] = old
    assert capi.host_addressing_fingerprint(case) == h0
    # interface cells and kinds count too
    c2 = synthetic.poisson_block(8, 8, 8, 2, 1, 1, 0)
    h2 = capi.host_addressing_fingerprint(c2)
    c2.interfaces[0].face_cells[3] ^= 1
    assert capi.host_addressing_fingerprint(c2) != h2


def test_rows_of_a_wavefront_are_put_longest_first_when_lengths_are_mixed():
    """Mixed cell types (a hex mesh that lost 30 % of its faces: rows of 1..7 entries): the lanes of the
    compressed SpMV stop loading at the end of their own rows, so padding costs traffic only where it shares
    a 128-byte line with a slot in use.  With `renumber` the rows of every wavefront (128 consecutive rows)
    go longest first: the lines in use are dense, the layout qualifies, and no row leaves its wavefront (the
    gather is as local as it was)."""
    case = synthetic.drop_faces_case(synthetic.poisson_case(28), 0.3)
    d0, loc0, _, _ = capi.host_pattern(case)
    rp0 = rowptr_of(loc0[0], d0.n_rows)
    assert capi.host_sell_read_slots(rp0, loc0[1])[0] is False       # as given: lines of mixed lengths
    rcm = capi.host_rcm(rp0, loc0[1])
    for mode in (capi.RENUMBER_ON, capi.RENUMBER_AUTO):
        d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(case, mode)
        assert ren and sorted(new_id.tolist()) == list(range(case.n_cells))
        # every row stays in the wavefront the numbering at large (RCM, or the caller's when `auto` keeps
        # it) put it in
        _, _, _, _, (ren0, base) = capi.host_pattern_renumbered(case, mode, compress_indices=0)
        base = base if ren0 else np.arange(case.n_cells)
        if mode == capi.RENUMBER_ON:
            np.testing.assert_array_equal(base, rcm)
        np.testing.assert_array_equal(new_id // 128, base // 128)
        rp = rowptr_of(loc[0], d.n_rows)
        ok, allocated, read = capi.host_sell_read_slots(rp, loc[1])
        assert ok and read <= 1.15 * d.local_nnz + 8 * 512 < allocated
        lens = np.diff(rp)
        for c in range(0, d.n_rows, 128):                            # longest first inside every wavefront
            assert np.all(np.diff(lens[c:c + 128]) <= 0)
    # a hex mesh has nothing to gain: the numbering is the plain RCM one
    box = synthetic.renumber_case(synthetic.poisson_case(28), 4096)
    d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(box, capi.RENUMBER_ON)
    d1, loc1, _, _ = capi.host_pattern(box)
    np.testing.assert_array_equal(new_id, capi.host_rcm(rowptr_of(loc1[0], d1.n_rows), loc1[1]))
    # without the compressed layout in play nothing is sorted
    d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(case, capi.RENUMBER_ON, compress_indices=0)
    np.testing.assert_array_equal(new_id, rcm)


def test_rows_are_not_sorted_where_the_slot_major_gather_is_not_local():
    """Mixed row lengths again, but neighbours picked at random within +-300 cells (and a Voronoi mesh): the
    s-th neighbours of the rows of a wavefront have nothing to do with each other, the compressed layout's
    slot-major gather loses against the CSR-stream kernel's row-major one -- the rows stay in plain RCM
    order (consecutive rows = neighbouring cells, what the CSR-stream kernel wants)."""
    for case in (synthetic.random_global_case(20000, 4, 300, symmetric=True, seed=2), synthetic.voronoi_case(20000)):
        d0, loc0, _, _ = capi.host_pattern(case)
        rcm = capi.host_rcm(rowptr_of(loc0[0], d0.n_rows), loc0[1])
        d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(case, capi.RENUMBER_ON)
        assert ren
        np.testing.assert_array_equal(new_id, rcm)
        assert capi.host_sell_read_slots(rowptr_of(loc[0], d.n_rows), loc[1])[0] is False


def test_hilbert_order_through_the_cell_centres_is_the_better_candidate_on_a_polyhedral_mesh():
    """ogl_ldu_view::cell_centres (this build's addition: the plug-in passes mesh.C()): the cells along a Hilbert curve
    through their centres gather x from fewer 64-byte sectors than reverse Cuthill-McKee does on a Voronoi mesh -- a
    chunk of consecutive rows is a compact blob instead of a strip of a breadth-first front -- and the policy takes
    the better of the two candidates; without centres nothing changes."""
    n = 30000
    case = synthetic.voronoi_case(n, with_centres=True)
    d0, loc0, _, _ = capi.host_pattern(case)
    rp, cols = rowptr_of(loc0[0], d0.n_rows), loc0[1]
    curve = capi.host_hilbert_order(case.centres)
    assert sorted(curve.tolist()) == list(range(n))
    r_nat = capi.host_gather_sector_ratio(rp, cols)
    r_rcm = capi.host_gather_sector_ratio(rp, cols, capi.host_rcm(rp, cols))
    r_cur = capi.host_gather_sector_ratio(rp, cols, curve)
    assert r_cur < 0.8 * r_rcm < 0.3 * r_nat, (r_nat, r_rcm, r_cur)
    # neighbours along the curve are neighbours in space: the mean distance between consecutive cells is a few cell sizes
    order = np.argsort(curve)
    step = np.linalg.norm(np.diff(case.centres[order], axis=0), axis=1)
    assert np.mean(step) < 3.0 * n ** (-1.0 / 3.0)
    d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(case, capi.RENUMBER_ON, compress_indices=0)
    assert ren
    np.testing.assert_array_equal(new_id, curve)
    import dataclasses
    d, loc, _, _, (ren, new_id) = capi.host_pattern_renumbered(dataclasses.replace(case, centres=None), capi.RENUMBER_ON,
                                                               compress_indices=0)
    np.testing.assert_array_equal(new_id, capi.host_rcm(rp, cols))
    # a structured box keeps its numbering whatever it is handed
    box = synthetic.poisson_case(20)
    i = np.arange(box.n_cells)
    box = dataclasses.replace(box, centres=np.stack([i % 20, (i // 20) % 20, i // 400], axis=1).astype(float))
    _, _, _, _, (ren, _) = capi.host_pattern_renumbered(box, capi.RENUMBER_AUTO)
    assert not ren
