"""Product host logic (libogl_amd.so, ogl_host_* entry points) on the CPU:
 - against the reference's known-answer vectors (unitTests/test_HostMatrix.C:8-107)
 - against the oracle on generated cases incl. cyclic and processor interfaces
 - the library loads without a GPU and exports every symbol include/ogl_amd.h declares
No compute entry point is called here (they need a gfx950 device)."""
import os
import re

import numpy as np
import pytest

from ogl_amd import capi, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    header = open(os.path.join(ROOT, "include", "ogl_amd.h")).read()
    declared = set(re.findall(r"\b(ogl_[a-z_0-9]+)\s*\(", header))
    declared -= {"ogl_allreduce_sum_fn", "ogl_neighbour_exchange_fn"}
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert lib.ogl_abi_version() == 4


def test_no_device_fails_loudly():
    import ctypes
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        has_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.OglError) as e:
        capi.Registry()
    assert e.value.status == capi.ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_config_defaults_are_the_codes_defaults():
    c = capi.default_config()
    assert (c.tolerance, c.rel_tol, c.max_iter, c.min_iter) == (1e-6, 1e-6, 1000, 0)
    assert (c.eval_frequency, c.norm_eval_limit, c.relaxation_factor, c.adapt_min_iter) == (1, 100, 0.6, 1)
    assert c.matrix_format == capi.FORMAT_COO and c.update_rhs == 1 and c.update_init_guess == 0
    assert c.update_sys_matrix == 1 and c.regenerate == 0 and c.scaling == 1.0
    assert c.max_block_size == 1 and c.caching == 0 and c.ranks_per_gpu == 1


# ---- the reference's gtest cases, through the product's free functions -------------------

def test_symmetric_update(golden):
    g = golden["symmetric_update"]
    res = capi.host_symmetric_update(g["permute"], g["scale"], g["diag"], g["upper"])
    assert res.tolist() == [float(v) for v in g["expected"]]
    res = capi.host_symmetric_update(g["permute"], -1.0, g["diag"], g["upper"])
    assert res.tolist() == [float(v) for v in g["expected"]]      # scale ignored


def test_non_symmetric_update(golden):
    g = golden["non_symmetric_update"]
    res = capi.host_non_symmetric_update(g["permute"], g["scale"], g["diag"], g["upper"], g["lower"])
    assert res.tolist() == [float(v) for v in g["expected"]]


@pytest.mark.parametrize("case", ["init_local_sparsity", "init_local_sparsity_asym",
                                  "init_local_sparsity_box2"])
def test_init_local_sparsity(golden, case):
    g = golden[case]
    rows, cols, perm = capi.host_init_local_sparsity(g["nrows"], g["upper"], g["lower"],
                                                     g["is_symmetric"])
    assert rows.tolist() == g["rows"]
    assert cols.tolist() == g["cols"]
    assert perm.tolist() == g["permute"]


# ---- against the oracle ---------------------------------------------------------------

def _orc_ifaces(oracle, case):
    return [oracle.Iface(f.kind, f.face_cells, f.bou_coeffs, f.neighb_proc, f.neighb_patch)
            for f in case.interfaces]


@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("shape,kw", [((6, 5, 4), {}), ((6, 5, 4), {"periodic_x": True}),
                                      ((4, 4, 8), {"pz": 2, "rank": 1}),
                                      ((6, 4, 4), {"px": 3, "rank": 1}),
                                      ((4, 4, 4), {"px": 2, "py": 2, "pz": 2, "rank": 5})])
def test_pattern_matches_oracle(oracle, sym, shape, kw):
    case = synthetic.poisson_block(*shape, symmetric=sym, off_upper=-0.9,
                                   off_lower=-0.9 if sym else -1.1, **kw)
    ifs = _orc_ifaces(oracle, case)
    d, loc, nl, comm = capi.host_pattern(case)
    rows, cols, perm = oracle.init_local_sparsity_pattern(case.n_cells, case.upper_addr,
                                                          case.lower_addr, sym, ifs)
    assert d.n_rows == case.n_cells and d.local_nnz == rows.size
    np.testing.assert_array_equal(loc[0], rows)
    np.testing.assert_array_equal(loc[1], cols)
    np.testing.assert_array_equal(loc[2], perm)
    r2, c2, p2 = oracle.init_non_local_sparsity(ifs)
    np.testing.assert_array_equal(nl[0], r2)
    np.testing.assert_array_equal(nl[1], c2)
    np.testing.assert_array_equal(nl[2], p2)
    ids, sizes, send = oracle.create_communication_pattern(ifs)
    np.testing.assert_array_equal(comm[0], ids)
    np.testing.assert_array_equal(comm[1], sizes)
    np.testing.assert_array_equal(comm[2], send)
    assert np.all(np.diff(comm[0]) > 0)                        # ascending neighbour rank


def test_unsorted_faces_are_sorted_like_the_reference(oracle):
    # the reference sorts by (row, col) whatever the face order (HostMatrixFreeFunctions.C:120-148)
    rng = np.random.default_rng(20241016)
    lo, up = synthetic.box_faces(4, 3, 3)
    perm = rng.permutation(lo.size)
    lo, up = lo[perm], up[perm]
    for sym in (True, False):
        a = capi.host_init_local_sparsity(36, up, lo, sym)
        b = oracle.init_local_sparsity(36, up, lo, sym)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_update_w_interface_matches_oracle(oracle):
    case = synthetic.poisson_block(5, 4, 3, symmetric=False, periodic_x=True, off_upper=-0.9,
                                   off_lower=-1.1)
    ifs = _orc_ifaces(oracle, case)
    _, loc, _, _ = capi.host_pattern(case)
    cc = oracle.collect_interface_coeffs(ifs, True)
    a = capi.host_non_symmetric_update_w_interface(loc[2], 2.0, case.diag, case.upper, case.lower, cc)
    b = oracle.non_symmetric_update_w_interface(loc[2], 2.0, case.diag, case.upper, case.lower, cc)
    np.testing.assert_array_equal(a, b)
    # the symmetric variant on the mapping of a symmetric pattern (positions: upper | diag | interfaces)
    case = synthetic.poisson_block(5, 4, 3, symmetric=True, periodic_x=True, off_upper=-0.9)
    ifs = _orc_ifaces(oracle, case)
    _, loc, _, _ = capi.host_pattern(case)
    cc = oracle.collect_interface_coeffs(ifs, True)
    assert loc[2].max() < case.upper.size + case.diag.size + cc.size
    a = capi.host_symmetric_update_w_interface(loc[2], 2.0, case.diag, case.upper, cc)
    b = oracle.symmetric_update_w_interface(loc[2], 2.0, case.diag, case.upper, cc)
    np.testing.assert_array_equal(a, b)


def test_invalid_views_are_rejected():
    case = synthetic.poisson_case(3)
    case.upper_addr = case.upper_addr.copy()
    case.upper_addr[0] = 99
    with pytest.raises(capi.OglError) as e:
        capi.host_pattern(case)
    assert e.value.status == capi.ERR_INVALID
    case = synthetic.poisson_block(4, 4, 4, periodic_x=True)
    case.interfaces[0].kind = 7            # e.g. cyclicAMI: FatalError in the reference
    with pytest.raises(capi.OglError) as e:
        capi.host_pattern(case)
    assert e.value.status == capi.ERR_UNSUPPORTED


def test_empty_matrix():
    case = synthetic.LduCase(0, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0),
                             np.zeros(0), None)
    d, loc, nl, comm = capi.host_pattern(case)
    assert (d.n_rows, d.local_nnz, d.non_local_nnz, d.n_neighbours) == (0, 0, 0, 0)


def test_adapt_criterion_matches_oracle(oracle):
    for prev_iters, cost, export in [(1, 0.0, 0), (100, 1.0, 0), (100, 1.0, 1), (37, 12.0, 0),
                                     (100000, 1.0, 0), (50, 400.0, 0)]:
        cfg = capi.default_config(export_res=export, min_iter=2, eval_frequency=3)
        got = capi.host_adapt_criterion(cfg, prev_iters, cost)
        exp = oracle.adapt_criterion(2, 3, export, prev_iters, True, 0.6, 100, cost)
        assert got == exp


def test_memory_ledger_reads_without_a_device():
    """ogl_memory_ledger_read is pure bookkeeping: with nothing allocated (no GPU here) every field is 0, and the struct
    has the layout include/ogl_amd.h declares (14 x int64)."""
    led = capi.memory_ledger()
    assert ctypes_sizeof(led) == 14 * 8
    d = led.as_dict()
    assert set(d) == {"device_bytes", "device_blocks", "device_peak_bytes", "device_alloc_calls", "pinned_bytes", "pinned_blocks",
                      "pinned_peak_bytes", "pinned_alloc_calls", "streams", "events", "graph_execs", "events_created",
                      "graph_execs_created", "unknown_frees"}
    if not _has_gpu():
        assert all(v == 0 for v in d.values()), d


def ctypes_sizeof(obj):
    import ctypes
    return ctypes.sizeof(obj)


def _has_gpu():
    import ctypes
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False
