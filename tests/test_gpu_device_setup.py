"""The once-per-pattern set-up on the device (setup_kernels.hip: lduMatrix addressing -> row-major pattern +
ldu_mapping by counting / scan / fill / per-row sort; half-storage mask and map from it) against the host
algorithm (host_matrix.cpp, pinned by the reference's gtest vectors in the CPU suite) and against the oracle:
integer arrays bit-exact, whatever path built them."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import oracle_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def cfg(**kw):
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-11, rel_tol=0.0, max_iter=40,
                export_res=1, matrix_format=capi.FORMAT_CSR, adapt_min_iter=0, renumber=capi.RENUMBER_OFF)
    base.update(kw)
    return capi.default_config(**base)


def randomise(case, seed):
    rng = np.random.default_rng(seed)
    case.upper[:] = rng.uniform(-1.0, -0.25, case.upper.size)
    if case.lower is not None:
        case.lower[:] = rng.uniform(-1.0, -0.25, case.lower.size)
    case.diag[:] = rng.uniform(7.0, 9.0, case.n_cells)
    return case


CASES = [
    ("box", lambda: synthetic.poisson_block(33, 31, 29)),
    ("box_asym", lambda: synthetic.poisson_block(21, 9, 14, symmetric=False, off_upper=-0.9, off_lower=-1.1)),
    ("periodic", lambda: synthetic.poisson_block(16, 15, 14, periodic_x=True)),
    ("periodic_asym", lambda: synthetic.poisson_block(6, 5, 4, periodic_x=True, symmetric=False, off_upper=-0.9,
                                                      off_lower=-1.1)),
    ("one_cell", lambda: synthetic.poisson_block(1, 1, 1)),
    ("line", lambda: synthetic.poisson_block(5000, 1, 1)),
    ("shuffled", lambda: synthetic.renumber_case(synthetic.poisson_case(24), 4096)),
    ("random", lambda: synthetic.random_global_case(3000, 3, 900, symmetric=False, seed=5)),
    ("long_rows", lambda: synthetic.long_rows_case(synthetic.poisson_case(20), 0.05, 20)),
    ("voronoi", lambda: synthetic.voronoi_case(6000)),
]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_device_built_pattern_equals_host_built_and_oracle(reg, oracle, name, make):
    case = randomise(make(), 11)
    dev = reg.solver("ds_dev_" + name, cfg())
    host = reg.solver("ds_host_" + name, cfg())
    host.set_property("deviceSetup", 0.0)
    dev.set_matrix(case)
    host.set_matrix(case)
    assert dev.get_property("patternBuiltOnDevice") == (1.0 if case.n_cells > 0 else 0.0)
    assert host.get_property("patternBuiltOnDevice") == 0.0
    rp, cols, vals = oracle_csr(oracle, case)
    a, b = dev.local_matrix(), host.local_matrix()
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[0], rp)
    np.testing.assert_array_equal(a[1], cols)
    np.testing.assert_array_equal(a[3], vals)
    assert dev.get_property("symmetricHalf") == host.get_property("symmetricHalf")
    x = np.random.default_rng(3).uniform(-1, 1, case.n_cells)
    ref = oracle.spmv(rp, cols, vals, x)
    np.testing.assert_array_equal(dev.spmv(x), ref)
    np.testing.assert_array_equal(host.spmv(x), ref)
    rhs = np.random.default_rng(4).uniform(-1, 1, case.n_cells)
    xd, pd = dev.solve(rhs, np.zeros_like(rhs))
    xh, ph = host.solve(rhs, np.zeros_like(rhs))
    np.testing.assert_array_equal(dev.history(), host.history())
    np.testing.assert_array_equal(xd, xh)


def test_device_set_up_feeds_every_host_consumer(reg, oracle):
    """Block-Jacobi blocks, the ISAI pattern, matrixFormat Ell, reorderOnHost and the renumbering policy read the
    pattern on the host: a device-built pattern is downloaded for them on demand."""
    case = randomise(synthetic.renumber_case(synthetic.poisson_case(14), 700), 2)
    rhs = np.random.default_rng(4).uniform(-1, 1, case.n_cells)
    variants = [dict(max_block_size=4), dict(preconditioner=capi.PRECOND_ISAI), dict(matrix_format=capi.FORMAT_ELL),
                dict(reorder_on_host=1), dict(renumber=capi.RENUMBER_ON), dict(compress_indices=0)]
    for i, kw in enumerate(variants):
        out = []
        for device in (1.0, 0.0):
            s = reg.solver(f"ds_cons_{i}_{int(device)}", cfg(**kw))
            s.set_property("deviceSetup", device)
            s.set_matrix(case)
            assert s.get_property("patternBuiltOnDevice") == device
            x, perf = s.solve(rhs, np.zeros_like(rhs))
            out.append((x, s.history().copy()))
        np.testing.assert_array_equal(out[0][1], out[1][1], err_msg=str(kw))
        np.testing.assert_array_equal(out[0][0], out[1][0], err_msg=str(kw))


def test_non_conforming_addressing_takes_the_host_path(reg):
    """A face with owner > neighbour is not lduAddressing; the host algorithm keeps the reference's segment order
    for it ([lower | diag | upper] by face role, HostMatrixFreeFunctions.C:105-201) and the device build steps aside."""
    case = randomise(synthetic.poisson_block(7, 6, 5, symmetric=False, off_upper=-0.9, off_lower=-1.1), 4)
    f = np.arange(0, case.n_faces, 7)
    case.lower_addr[f], case.upper_addr[f] = case.upper_addr[f].copy(), case.lower_addr[f].copy()
    s = reg.solver("ds_nonconf", cfg()).set_matrix(case)
    assert s.get_property("patternBuiltOnDevice") == 0.0 and s.get_property("deviceSetup") == 1.0
    h = reg.solver("ds_nonconf_h", cfg())
    h.set_property("deviceSetup", 0.0)
    h.set_matrix(case)
    for x, y in zip(s.local_matrix(), h.local_matrix()):
        np.testing.assert_array_equal(x, y)
    # and it is the operator the faces describe
    A = np.zeros((case.n_cells, case.n_cells))
    A[np.arange(case.n_cells), np.arange(case.n_cells)] = case.diag
    A[case.lower_addr, case.upper_addr] = case.upper
    A[case.upper_addr, case.lower_addr] = case.lower
    x = np.random.default_rng(8).uniform(-1, 1, case.n_cells)
    np.testing.assert_allclose(s.spmv(x), A @ x, rtol=1e-13, atol=1e-13)


def test_face_outside_the_mesh_is_refused(reg):
    case = synthetic.poisson_case(6)
    case.upper_addr[17] = case.n_cells
    with pytest.raises(capi.OglError):
        reg.solver("ds_bad", cfg()).set_matrix(case)
    case.upper_addr[17] = -3
    with pytest.raises(capi.OglError):
        reg.solver("ds_bad2", cfg()).set_matrix(case)


def test_pattern_rebuilds_follow_the_addressing(reg, oracle):
    s = reg.solver("ds_rebuild", cfg())
    rng = np.random.default_rng(1)
    for make in (lambda: synthetic.poisson_case(9), lambda: synthetic.poisson_case(12),
                 lambda: synthetic.renumber_case(synthetic.poisson_case(12), 64),
                 lambda: synthetic.poisson_block(12, 12, 12, periodic_x=True), lambda: synthetic.poisson_case(12)):
        case = randomise(make(), 6)
        s.set_matrix(case)
        rp, cols, vals = oracle_csr(oracle, case)
        x = rng.uniform(-1, 1, case.n_cells)
        np.testing.assert_array_equal(s.spmv(x), oracle.spmv(rp, cols, vals, x))


GRAPHS = [
    ("shuffled", lambda: synthetic.renumber_case(synthetic.poisson_case(24), 4096)),
    ("box", lambda: synthetic.poisson_block(33, 31, 29)),
    ("periodic", lambda: synthetic.poisson_block(16, 15, 14, periodic_x=True)),
    ("random", lambda: synthetic.random_global_case(3000, 3, 900, symmetric=False, seed=5)),
    ("components", lambda: synthetic.drop_faces_case(synthetic.poisson_block(60, 4, 1), 0.45)),
    ("voronoi", lambda: synthetic.voronoi_case(20000)),
    ("long_rows", lambda: synthetic.renumber_case(synthetic.long_rows_case(synthetic.poisson_case(20), 0.05, 20), 512)),
]


@pytest.mark.parametrize("name,make", GRAPHS, ids=[g[0] for g in GRAPHS])
def test_device_rcm_and_renumbering_equal_the_host_algorithms(reg, oracle, name, make):
    """renumber on, compressed layout off: the numbering in use is the plain reverse Cuthill-McKee order.  Built
    level by level on the device it must be the host's rcm_order (ogl_host_rcm) node for node, and the pattern
    rewritten on the device the host's renumber_pattern."""
    case = randomise(make(), 13)
    rp, cols, vals = oracle_csr(oracle, case)
    out = {}
    for device in (1.0, 0.0):
        s = reg.solver(f"ds_rcm_{name}_{int(device)}", cfg(renumber=capi.RENUMBER_ON, compress_indices=0))
        s.set_property("deviceSetup", device)
        s.set_matrix(case)
        assert s.get_property("patternBuiltOnDevice") == device
        assert s.get_property("renumberedOnDevice") == device
        out[device] = (s.renumbering(), s.local_matrix())
    np.testing.assert_array_equal(out[1.0][0], capi.host_rcm(rp, cols))
    np.testing.assert_array_equal(out[1.0][0], out[0.0][0])
    for a, b in zip(out[1.0][1], out[0.0][1]):
        np.testing.assert_array_equal(a, b)
    # and it is the oracle's matrix permuted by that numbering
    p_rp, p_cols, p_vals, _ = oracle.permute_csr(rp, cols, vals, out[1.0][0])
    np.testing.assert_array_equal(out[1.0][1][0], p_rp)
    np.testing.assert_array_equal(out[1.0][1][1], p_cols)
    np.testing.assert_array_equal(out[1.0][1][3], p_vals)


def test_device_rcm_leaves_chains_to_the_host(reg):
    """A graph with tens of thousands of breadth-first levels is no work for a level-synchronous search."""
    case = synthetic.poisson_block(70000, 1, 1)
    s = reg.solver("ds_chain", cfg(renumber=capi.RENUMBER_ON, compress_indices=0)).set_matrix(case)
    rp, cols, _, _ = s.local_matrix()
    assert s.renumbering() is not None
    assert s.get_property("patternBuiltOnDevice") == 1.0
