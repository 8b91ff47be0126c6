"""Half storage of a symmetric matrix on a banded pattern (config symmetric_half; build_sym_layout in
ogl_amd/csrc/host_matrix.cpp): ogl_host_sym_check builds the layout and walks every row the way k_spmv_sym
does -- lower entries read where their upper twins live -- and fails if that does not reproduce the pattern."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic


def rowptr_of(rows, n):
    return np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)


def pattern(case):
    d, loc, _, _ = capi.host_pattern(case)
    return rowptr_of(loc[0], d.n_rows), loc[1], d


def test_box_takes_four_planes():
    rp, cols, d = pattern(synthetic.poisson_case(20))
    ok, dist, slots, used = capi.host_sym_check(rp, cols)
    assert ok and dist == [0, 1, 20, 400]
    n_chunks = (d.n_rows + 511) // 512
    assert slots == 4 * 512 * n_chunks
    assert used == d.n_rows + (d.local_nnz - d.n_rows) // 2          # diagonal + one of every pair


def test_slabs_lines_and_sheets():
    for kw, dist in ((dict(gx=16, gy=16, gz=32, pz=2, rank=1), [0, 1, 16, 256]),     # a rank's slab: same structure
                     (dict(gx=1031, gy=1, gz=1), [0, 1]),                           # 1-D: tridiagonal
                     (dict(gx=64, gy=64, gz=1), [0, 1, 64]),                        # 2-D: 5-point
                     (dict(gx=33, gy=31, gz=29), [0, 1, 33, 33 * 31]),              # partial last chunk
                     (dict(gx=5, gy=4, gz=3), [0, 1, 5, 20])):
        rp, cols, _ = pattern(synthetic.poisson_block(**kw))
        ok, got, _, _ = capi.host_sym_check(rp, cols)
        assert ok and got == dist, kw


def test_patterns_that_do_not_qualify():
    # a cyclic pair adds a fifth distance
    rp, cols, _ = pattern(synthetic.poisson_block(6, 5, 4, periodic_x=True))
    assert capi.host_sym_check(rp, cols)[0] is False
    # a shuffled numbering has no bands
    rp, cols, _ = pattern(synthetic.renumber_case(synthetic.poisson_case(16), 512))
    assert capi.host_sym_check(rp, cols)[0] is False
    # a lower entry without its upper twin (structurally non-symmetric)
    n = 1024
    rows = [[r] + ([r + 1] if r + 1 < n else []) + ([r - 1] if r % 3 == 0 and r > 0 else []) for r in range(n)]
    rows = [np.array(sorted(c)) for c in rows]
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    # (r, r-1) exists for r % 3 == 0 and (r-1, r) exists for every r: twins are there -> qualifies ...
    assert capi.host_sym_check(rp, np.concatenate(rows).astype(np.int32))[0] is True
    # ... but not the other way round: (r, r-1) everywhere, (r-1, r) only sometimes
    rows = [[r] + ([r - 1] if r > 0 else []) + ([r + 1] if r % 3 == 0 and r + 1 < n else []) for r in range(n)]
    rows = [np.array(sorted(c)) for c in rows]
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    assert capi.host_sym_check(rp, np.concatenate(rows).astype(np.int32))[0] is False
    # too much padding: a box that lost 30 % of its faces fills 4 planes with 3.1 entries per row
    rp, cols, _ = pattern(synthetic.drop_faces_case(synthetic.poisson_case(28), 0.3))
    assert capi.host_sym_check(rp, cols)[0] is False
    # an octree mesh: more distances than planes
    rp, cols, _ = pattern(synthetic.octree_case(16, 1.5))
    assert capi.host_sym_check(rp, cols)[0] is False


# ---- half storage with per-chunk distances and explicit exceptions (build_symx_layout) ----
def _rowptr(rows, n):
    return np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)


SYMX_CASES = [
    ("two_blocks", lambda: synthetic.multi_block_case([30, 17], 24, 20), True),
    ("three_blocks", lambda: synthetic.multi_block_case([12, 40, 9], 16, 33), True),
    ("thin_blocks", lambda: synthetic.multi_block_case([3, 5, 2, 7], 9, 11), None),   # (blocks shorter than a chunk)
    ("box", lambda: synthetic.poisson_case(20), True),
    ("line", lambda: synthetic.poisson_block(1000, 1, 1), True),
    ("one_cell", lambda: synthetic.poisson_block(1, 1, 1), True),
    ("periodic", lambda: synthetic.poisson_block(16, 15, 14, periodic_x=True), None),   # (cyclic entries: explicit)
    ("shuffled", lambda: synthetic.renumber_case(synthetic.poisson_case(16), 512), False),
    ("voronoi", lambda: synthetic.voronoi_case(4000), False),
    ("octree", lambda: synthetic.octree_case(24, 1.5), None),
]


@pytest.mark.parametrize("name,make,qualifies", SYMX_CASES, ids=[c[0] for c in SYMX_CASES])
def test_symx_layout_decodes_to_the_pattern(name, make, qualifies):
    """ogl_host_symx_check builds the layout and walks every row as k_spmv_symx does (lower entries from their
    twins' planes, explicit entries merged by column): it must reproduce the row-major pattern entry by entry,
    whether the layout is worth using or not."""
    case = make()
    d, loc, _, _ = capi.host_pattern(case)
    rp = _rowptr(loc[0], d.n_rows)
    ok, slots, planar, explicit, ex_chunks, chunks = capi.host_symx_check(rp, loc[1])
    assert planar + explicit == d.local_nnz
    assert chunks == -(-d.n_rows // 512) and slots % 512 == 0
    if qualifies is not None:
        assert ok == qualifies, (name, planar / d.local_nnz)
    if name in ("box", "line", "one_cell"):
        assert explicit <= 0.01 * d.local_nnz + 8
    if name in ("two_blocks", "three_blocks"):
        assert planar >= 0.9 * d.local_nnz and explicit > 0


def test_symx_kernel_split_and_pair_shape():
    """Which kernel instantiation the per-chunk half storage ends up on (host side of the decision): blocks with even
    line lengths stay on the pair-load instantiation even though the chunks at their seams see odd distances (those
    keep only pair-addressable ones); the lean kernel takes every chunk whose rows have at most one explicit entry
    ahead of and one behind their planar ones -- block faces -- and the general kernel the rest."""
    def kernels(case):
        d, loc, _, _ = capi.host_pattern(case)
        rp = _rowptr(loc[0], d.n_rows)
        ok = capi.host_symx_check(rp, loc[1])
        return ok, capi.host_symx_kernels(rp, loc[1])

    # even line lengths everywhere: pair-shaped, and only the seam chunks are general
    ok, (fast, general) = kernels(synthetic.multi_block_case([60, 40], 48, 30))
    assert ok[0] and fast and 0 < general <= 8 and ok[4] == ok[5]        # (every chunk has explicit entries)
    # an odd line length in one block: that block's chunks cannot pair their rows
    ok, (fast, general) = kernels(synthetic.multi_block_case([30, 17], 24, 20))
    assert ok[0] and not fast and general <= 2
    # a plain box: nothing for the general kernel (the last, partly filled chunk keeps a few entries explicit)
    ok, (fast, general) = kernels(synthetic.poisson_case(20))
    assert ok[0] and fast and general == 0 and ok[3] <= 0.01 * (ok[2] + ok[3])
    # random couplings inside a band: rows with several explicit entries between their planar ones
    ok, (fast, general) = kernels(synthetic.random_global_case(1029, 3, 4, seed=8))
    assert ok[0] and general > 0
