"""The synthetic stand-ins for unstructured meshes that bench.py offers (ogl_amd/synthetic.py): they must be
valid lduMatrix addressing (owner < neighbour, upper-triangular order, every face once), and the host-side
policy must treat them as DESIGN.md says."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic


def rowptr_of(rows, n):
    return np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)


def check_addressing(case):
    lo, up = case.lower_addr.astype(np.int64), case.upper_addr.astype(np.int64)
    assert np.all(lo < up) and up.max() < case.n_cells
    key = lo * case.n_cells + up
    assert np.all(np.diff(key) > 0)                      # upper-triangular order, no face twice
    # diag = #faces + delta and upper = -1: A 1 = delta
    y = synthetic.apply_case(case, np.ones(case.n_cells))
    np.testing.assert_allclose(y, 1e-3 * (1.0 + (case.global_index % 7) / 7.0), rtol=0, atol=1e-13)


@pytest.mark.parametrize("append", [False, True])
def test_octree_case_is_a_hex_dominant_mesh(append):
    n = 24
    case = synthetic.octree_case(n, 1.5, append)
    check_addressing(case)
    d, loc, _, _ = capi.host_pattern(case)
    lens = np.bincount(loc[0], minlength=d.n_rows)
    # bulk rows have 7 entries (fewer on the box's walls); an unsplit cell next to a split one has 4 faces on
    # that side: 10, 13 or 16 entries
    assert set(np.unique(lens)) <= {4, 5, 6, 7, 10, 13, 16} and (lens == 10).sum() > 0
    n_split = (case.n_cells - n ** 3) // 7
    assert case.n_cells == n ** 3 + 7 * n_split and 0 < n_split < n ** 3 // 4
    # both numberings describe the same mesh: same multiset of row lengths
    other = synthetic.octree_case(n, 1.5, not append)
    d2, loc2, _, _ = capi.host_pattern(other)
    np.testing.assert_array_equal(np.sort(lens), np.sort(np.bincount(loc2[0], minlength=d2.n_rows)))


def test_octree_case_runs_on_the_compressed_layout_with_a_spill():
    case = synthetic.octree_case(32, 1.5)
    d, loc, _, _, (ren, _) = capi.host_pattern_renumbered(case, capi.RENUMBER_AUTO)
    rp = rowptr_of(loc[0], d.n_rows)
    ok, read, spilled = capi.host_sell_spilled(rp, loc[1])
    # the few long rows sit together on the shell's surfaces: their tails are spilled or cost a few lines,
    # nothing like padding every row to 16
    assert ok and read + 4 * spilled < 1.15 * d.local_nnz + 8 * 512
    assert not ren                                       # the numbering gathers well as it is


def test_voronoi_case_is_valid_addressing():
    case = synthetic.voronoi_case(3000)
    check_addressing(case)
    d, loc, _, _ = capi.host_pattern(case)
    lens = np.bincount(loc[0], minlength=d.n_rows)
    assert 14 < lens.mean() < 18 and lens.min() >= 5
