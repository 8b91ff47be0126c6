"""Property test (hypothesis) of the integer work on the path: for random small lduMatrix systems --
random upper-triangular face sets, symmetric or not, processor interfaces to random neighbour ranks
and cyclic patch pairs -- the product's host pattern (ogl_host_pattern) must equal the oracle's
restatement of HostMatrixFreeFunctions.C:105-201 / HostMatrix.C:251-586 entry for entry, and the
compressed layout must decode back to the pattern.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

from ogl_amd import capi, synthetic
from helpers import orc_ifaces


@st.composite
def ldu_cases(draw):
    n = draw(st.integers(1, 40))
    pairs = sorted(draw(st.sets(st.tuples(st.integers(0, n - 1), st.integers(0, n - 1)).filter(
        lambda p: p[0] < p[1]), max_size=4 * n)))
    lower = np.array([p[0] for p in pairs], dtype=np.int32)
    upper = np.array([p[1] for p in pairs], dtype=np.int32)
    f = len(pairs)
    rng = np.random.default_rng(draw(st.integers(0, 2 ** 31)))
    sym = draw(st.booleans())
    ifaces = []
    for nb in sorted(draw(st.sets(st.integers(0, 6), max_size=3))):      # ascending neighbour rank
        cells = np.array(draw(st.lists(st.integers(0, n - 1), min_size=1, max_size=6)), dtype=np.int32)
        ifaces.append(synthetic.Interface(synthetic.IFACE_PROCESSOR, cells, rng.uniform(-1, 1, cells.size),
                                          nb, -1))
    if draw(st.booleans()):                                               # one cyclic patch pair
        m = draw(st.integers(1, min(5, n)))
        a = np.array(draw(st.lists(st.integers(0, n - 1), min_size=m, max_size=m)), dtype=np.int32)
        b = np.array(draw(st.lists(st.integers(0, n - 1), min_size=m, max_size=m)), dtype=np.int32)
        p0 = len(ifaces)
        ifaces.append(synthetic.Interface(synthetic.IFACE_CYCLIC, a, rng.uniform(-1, 1, m), -1, p0 + 1))
        ifaces.append(synthetic.Interface(synthetic.IFACE_CYCLIC, b, rng.uniform(-1, 1, m), -1, p0))
    return synthetic.LduCase(n, lower, upper, rng.uniform(1, 2, n), rng.uniform(-1, 1, f),
                             None if sym else rng.uniform(-1, 1, f), ifaces)


@settings(max_examples=150, deadline=None)
@given(ldu_cases())
def test_host_pattern_equals_oracle_on_random_systems(oracle, case):
    ifs = orc_ifaces(oracle, case)
    rows, cols, perm = oracle.init_local_sparsity_pattern(case.n_cells, case.upper_addr, case.lower_addr,
                                                          case.symmetric, ifs)
    nl_rows, nl_cols, nl_perm = oracle.init_non_local_sparsity(ifs)
    ids, sizes, send_idxs = oracle.create_communication_pattern(ifs)
    d, loc, nl, comm = capi.host_pattern(case)
    assert d.n_rows == case.n_cells and d.local_nnz == rows.size and d.non_local_nnz == nl_rows.size
    for got, want in zip(loc, (rows, cols, perm)):
        np.testing.assert_array_equal(got, want)
    for got, want in zip(nl, (nl_rows, nl_cols, nl_perm)):
        np.testing.assert_array_equal(got, want)
    for got, want in zip(comm, (ids, sizes, send_idxs)):
        np.testing.assert_array_equal(got, want)
    rp = oracle.rowptr_from_rows(case.n_cells, rows)
    ok, slots, _, _ = capi.host_sell_check(rp, cols)
    assert (ok and slots >= rows.size) or not ok      # decodes back, or is refused: never wrong
