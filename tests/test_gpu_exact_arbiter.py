"""north_star's "residual history matching the reference to 1e-12 rel", settled with an exact arbiter (VERDICT r4
item 3).  Three histories of the same GKOCG + BJ solve:

  gpu    the HIP path (fixed reduction tree, device_common.hpp:10-14)
  seq    the oracle in the reference executor's left-to-right order ([UPSTREAM] Ginkgo reference kernels)
  exact  the oracle with every dot / norm1 / sum / SpMV row sum accumulated by error-free transformations and rounded
         once (ORC_REDUCE_EXACT, pinned against rational arithmetic in tests/test_oracle_exact.py)

Asserted per check k, on the running maxima of the relative deviations over checks 0..k:

  |gpu - exact|  <=  C * max(|seq - exact|, 1e-13),   C = 8

i.e. the device order is never meaningfully FURTHER from the exact history than the reference's own order is -- and
while the residual is above 1e-3 of its start it is closer by one to two orders of magnitude (a tree of pairwise sums
carries ~log2(n) eps, a left-to-right sum ~sqrt(n) eps).  Later in the solve CG has amplified the rounding noise of
EITHER order to the same 1e-11 .. 1e-10: a reference executor and an OMP executor of Ginkgo differ from each other by
as much.  The numbers land in profiles/r05_parity_deviation.txt (tools/parity_deviation.py).
Reference: StoppingCriterion/StoppingCriterion.C:92-113.
"""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import oracle_matrix

pytestmark = pytest.mark.gpu

C_ARBITER = 8.0


@pytest.fixture(scope="module")
def reg():
    r = capi.Registry()
    yield r
    r.close()


def three_histories(reg, oracle, edge, name, **kw):
    case = synthetic.poisson_case(edge)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    b, _ = synthetic.rhs_for_x_star(case)
    cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, export_res=1, adapt_min_iter=0,
                              matrix_format=capi.FORMAT_CSR, **kw)
    s = reg.solver(name, cfg).set_matrix(case)
    _, perf = s.solve(b, np.zeros_like(b))
    gpu = s.history()
    inv = oracle.jacobi_generate_scalar(rp, cols, vals)
    seq = oracle.cg(A, b, np.zeros_like(b), inv, **kw).history
    oracle.set_reduction(oracle.REDUCE_EXACT)
    try:
        ex = oracle.cg(A, b, np.zeros_like(b), inv, **kw).history
    finally:
        oracle.set_reduction(oracle.REDUCE_SEQUENTIAL)
    return gpu, seq, ex


def check(gpu, seq, ex, label):
    assert gpu.size == seq.size == ex.size, (gpu.size, seq.size, ex.size)
    d_gpu = np.maximum.accumulate(np.abs(gpu - ex) / ex)
    d_seq = np.maximum.accumulate(np.abs(seq - ex) / ex)
    assert (d_gpu <= C_ARBITER * np.maximum(d_seq, 1e-13)).all(), (d_gpu / np.maximum(d_seq, 1e-13)).max()
    rel = ex / ex[0]
    early = rel > 1e-3
    print(f"{label}: {gpu.size} checks; residual above 1e-3 of its start ({int(early.sum())} checks): |seq - exact| "
          f"{d_seq[early].max():.2e}, |gpu - exact| {d_gpu[early].max():.2e}; all checks: {d_seq[-1]:.2e}, {d_gpu[-1]:.2e}")
    return d_gpu, d_seq, early


def test_64_to_convergence(reg, oracle):
    gpu, seq, ex = three_histories(reg, oracle, 64, "arb64", tolerance=1e-9, rel_tol=0.0, max_iter=2000)
    d_gpu, d_seq, early = check(gpu, seq, ex, "64^3 GKOCG + BJ to 1e-9")
    assert ex[-1] < 1e-9 <= ex[-2]
    # above 1e-3 of the start: the device tree is the closer order, and BOTH meet 1e-12 against the exact history
    assert d_gpu[early].max() <= 0.1 * d_seq[early].max()
    assert d_gpu[early].max() <= 1e-14 and d_seq[early].max() <= 1e-12


def test_216_fifty_turns(reg, oracle):
    gpu, seq, ex = three_histories(reg, oracle, 216, "arb216", tolerance=0.0, rel_tol=0.0, max_iter=50)
    d_gpu, d_seq, _ = check(gpu, seq, ex, "216^3 GKOCG + BJ, 50 turns")
    # 10 M-term sums (measured, profiles/r05_parity_deviation.txt): the left-to-right order is 2.6e-12 from the exact
    # history after 50 turns, the device tree 2e-15 -- the 2.6e-12 between the device and the sequential oracle
    # (tests/test_gpu_fullsize_oracle.py) is the REFERENCE order's rounding, not the device's.  north_star's 1e-12 is met
    # by the device against the exact history with three orders of magnitude to spare
    assert d_gpu[-1] <= 1e-14 and d_gpu[-1] <= 0.01 * d_seq[-1]
