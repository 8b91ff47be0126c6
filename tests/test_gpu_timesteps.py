"""A time-step loop as OpenFOAM runs it -- per step and field: constructor (lookup-or-create by name), new coefficients,
solve (lduLduBase.H:189-308, HostMatrix.C:15-96) -- leaves nothing behind: the device memory in use and the host's
resident set stay where they were after the first steps, for every solver / preconditioner pair, with the persistent
pattern reused (`regenerate` off) and the preconditioner regenerated every step (caching 0)."""
import resource

import numpy as np
import pytest

from ogl_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def test_time_steps_leave_no_memory_behind():
    torch = pytest.importorskip("torch")
    reg = capi.Registry()
    sym, asym = synthetic.poisson_case(24), synthetic.poisson_case(24, symmetric=False)
    fields = []
    for i, (sk, pc, k, case) in enumerate([(capi.SOLVER_CG, capi.PRECOND_BJ, 1, sym), (capi.SOLVER_CG, capi.PRECOND_ISAI, 1, sym),
                                           (capi.SOLVER_BICGSTAB, capi.PRECOND_GISAI, 1, asym),
                                           (capi.SOLVER_GMRES, capi.PRECOND_BJ, 4, asym)]):
        cfg = capi.default_config(solver=sk, preconditioner=pc, max_block_size=k, tolerance=1e-6, rel_tol=0.0, max_iter=400,
                                  krylov_dim=20, update_init_guess=1)   # (psi re-uploaded: every step does a whole solve)
        fields.append((f"field{i}", cfg, case, synthetic.rhs_for_x_star(case)[0]))

    def in_use():
        free, total = torch.cuda.mem_get_info(0)
        return (total - free) / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3

    marks = {}
    for step in range(161):
        for name, cfg, case, b in fields:
            case.diag[:] = case.diag * (1.0 + 1e-9)           # the coefficients of this time step
            s = reg.solver(name, cfg).set_matrix(case)         # constructor of this step: lookup-or-create by field name
            x, perf = s.solve(b, np.zeros_like(b))
            assert 1 <= perf.n_iterations < 400 and perf.final_residual < 1e-6, (name, step, perf.n_iterations, perf.final_residual)
        if step in (20, 160):
            marks[step] = in_use()
    reg.close()
    (dev0, rss0), (dev1, rss1) = marks[20], marks[160]
    assert abs(dev1 - dev0) < 1.0, (marks, "device memory in use moved over 140 time steps")
    assert rss1 - rss0 < 16.0, (marks, "host resident set grew over 140 time steps")


def test_registries_come_and_go_without_residue():
    """A registry per run (objectRegistry analogue, DevicePersistent/Base/Base.H:53-137): creating one, solving on it and
    closing it -- stream, events, pinned staging buffers, every field's device arrays -- gives everything back."""
    torch = pytest.importorskip("torch")
    case = synthetic.poisson_case(20)
    b = synthetic.rhs_for_x_star(case)[0]

    def in_use():
        free, total = torch.cuda.mem_get_info(0)
        return (total - free) / 1e6

    marks = {}
    for i in range(41):
        reg = capi.Registry()
        for name, pc in (("p", capi.PRECOND_BJ), ("q", capi.PRECOND_ISAI)):
            s = reg.solver(name, capi.default_config(preconditioner=pc, tolerance=1e-8, rel_tol=0.0, max_iter=200)).set_matrix(case)
            x, perf = s.solve(b, np.zeros_like(b))
            assert perf.final_residual < 1e-8
        reg.close()
        if i in (5, 40):
            marks[i] = in_use()
    assert abs(marks[40] - marks[5]) < 1.0, marks
