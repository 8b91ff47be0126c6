"""ogl_solver_set_matrix_like: the components of a vector field share ONE lduMatrix (fvMatrix::solveSegregated) -- the
second and third component take the first one's device copy of upper / lower instead of uploading them again
(include/ogl_amd.h; the reference uploads everything per solver object, HostMatrix/HostMatrix.C:644-682,
lduLduBase/lduLduBase.H:224-237).  Whatever the library decides, the device matrix must be the oracle's."""
import dataclasses

import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import blocked, oracle_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def cfg(**kw):
    base = dict(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_BJ, tolerance=1e-10, rel_tol=0.0, max_iter=300,
                export_res=1, adapt_min_iter=0, matrix_format=capi.FORMAT_CSR)
    base.update(kw)
    return capi.default_config(**base)


def components(case):
    """Three views of one matrix: the same upper / lower ARRAYS, a diagonal of their own (addBoundaryDiag)."""
    out = []
    for k in range(3):
        c = dataclasses.replace(case, diag=case.diag + 0.01 * k * (1 + np.arange(case.n_cells) % 5))
        out.append((c, capi.LduArrays(c)))
    assert out[0][1].upper.ctypes.data == out[1][1].upper.ctypes.data == out[2][1].upper.ctypes.data
    return out


@pytest.mark.parametrize("sym", [False, True], ids=["asym", "sym"])
@pytest.mark.parametrize("shape", [(20, 17, 13), (64, 64, 40)])
def test_second_and_third_component_take_the_device_copy(reg, oracle, sym, shape):
    case = synthetic.poisson_block(*shape) if sym else synthetic.poisson_block(*shape, symmetric=False, off_upper=-0.9,
                                                                                 off_lower=-1.1)
    comp = components(case)
    kind = dict(solver=capi.SOLVER_CG) if sym else {}
    tag = f"{'s' if sym else 'a'}{shape[0]}"
    sv = [reg.solver(f"U{tag}{'xyz'[k]}", cfg(**kind)) for k in range(3)]
    sv[0].set_matrix(comp[0][1])
    sv[1].set_matrix(comp[1][1], like=sv[0])
    sv[2].set_matrix(comp[2][1], like=sv[1])                      # (a taker can be a donor: its copy is the same)
    assert [s.get_property("offDiagReused") for s in sv] == [0.0, 1.0, 1.0]
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, case.n_cells)
    for k in range(3):
        _, (rp, cols, vals) = oracle_matrix(oracle, comp[k][0])
        d_rp, d_cols, _, d_vals = sv[k].local_matrix()
        np.testing.assert_array_equal(d_cols, cols)
        np.testing.assert_array_equal(d_vals, vals)
        full = reg.solver(f"V{tag}{k}", cfg(**kind)).set_matrix(comp[k][1])
        assert full.get_property("offDiagReused") == 0.0
        x1, p1 = sv[k].solve(b, np.zeros_like(b))
        x0, p0 = full.solve(b, np.zeros_like(b))
        np.testing.assert_array_equal(x1, x0)
        np.testing.assert_array_equal(sv[k].history(), full.history())
    # the values-only refresh of the next time step (same arrays, new numbers): reuse again
    comp[0][1].upper[:] *= 1.25
    sv[0].set_matrix(comp[0][1])
    sv[1].set_matrix(comp[1][1], like=sv[0])
    assert sv[1].get_property("offDiagReused") == 1.0
    _, (rp, cols, vals) = oracle_matrix(oracle, dataclasses.replace(comp[1][0], upper=comp[0][1].upper))
    np.testing.assert_array_equal(sv[1].local_matrix()[3], vals)


def test_anything_that_does_not_fit_is_a_full_upload(reg, oracle):
    case = synthetic.poisson_block(16, 15, 14, symmetric=False, off_upper=-0.9, off_lower=-1.1)
    comp = components(case)
    a = reg.solver("Wx", cfg()).set_matrix(comp[0][1])
    # other host arrays with the same numbers: pointer identity fails
    other = capi.LduArrays(dataclasses.replace(comp[1][0], upper=comp[1][0].upper.copy(), lower=comp[1][0].lower.copy()))
    b = reg.solver("Wy", cfg()).set_matrix(other, like=a)
    assert b.get_property("offDiagReused") == 0.0
    # the donor's arrays were written to since it uploaded them: the sampled checksum disagrees
    comp[0][1].upper[0] *= 2.0
    c = reg.solver("Wz", cfg()).set_matrix(comp[2][1], like=a)
    assert c.get_property("offDiagReused") == 0.0
    _, (rp, cols, vals) = oracle_matrix(oracle, dataclasses.replace(comp[2][0], upper=comp[2][1].upper))
    np.testing.assert_array_equal(c.local_matrix()[3], vals)
    # another addressing (same sizes: the cells renamed)
    ren = synthetic.renumber_case(case, 64)
    d = reg.solver("Wq", cfg()).set_matrix(capi.LduArrays(ren), like=a)
    assert d.get_property("offDiagReused") == 0.0
    # reorderOnHost: d_source is never filled, nothing to take (and nothing to give)
    e = reg.solver("We", cfg(reorder_on_host=1)).set_matrix(comp[1][1], like=a)
    assert e.get_property("offDiagReused") == 0.0
    f = reg.solver("Wf", cfg()).set_matrix(comp[1][1], like=e)
    assert f.get_property("offDiagReused") == 0.0
    # a solver of another registry is no donor
    r2 = capi.Registry()
    try:
        g = r2.solver("Wx", cfg()).set_matrix(comp[0][1])
        h = reg.solver("Wh", cfg()).set_matrix(comp[1][1], like=g)
        assert h.get_property("offDiagReused") == 0.0
    finally:
        r2.close()
