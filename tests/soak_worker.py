"""Time-step soak in a process of its own (no torch, nothing else on the device from this process): per step and field the
constructor (lookup-or-create by name), new coefficients, a whole solve -- what OpenFOAM does with the plug-in every time
step (lduLduBase.H:189-308, HostMatrix.C:15-96; device objects created once per field and updated in place afterwards,
DevicePersistent/Base/Base.H:53-137, HostMatrix.C:79-95).  Prints one JSON line per mark -- the library's allocation
ledger (exact), the driver's view of the device (hipMemGetInfo: moves with the runtime's own pools) and the resident set
-- and a summary line with the drift of the former and the least-squares slope of the latter.

    python tests/soak_worker.py [steps=300] [mark_every=50] [size=24]
"""
import json
import resource
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from ogl_amd import capi, synthetic  # noqa: E402


def fields_of(size):
    sym, asym = synthetic.poisson_case(size), synthetic.poisson_case(size, symmetric=False)
    out = []
    for i, (sk, pc, k, case) in enumerate([(capi.SOLVER_CG, capi.PRECOND_BJ, 1, sym), (capi.SOLVER_CG, capi.PRECOND_ISAI, 1, sym),
                                           (capi.SOLVER_BICGSTAB, capi.PRECOND_GISAI, 1, asym),
                                           (capi.SOLVER_GMRES, capi.PRECOND_BJ, 4, asym)]):
        cfg = capi.default_config(solver=sk, preconditioner=pc, max_block_size=k, tolerance=1e-6, rel_tol=0.0, max_iter=400,
                                  krylov_dim=20, update_init_guess=1)   # (psi re-uploaded: every step does a whole solve)
        out.append((f"field{i}", cfg, case, synthetic.rhs_for_x_star(case)[0]))
    return out


def one_step(reg, fields, step):
    for name, cfg, case, b in fields:
        case.diag[:] = case.diag * (1.0 + 1e-9)           # the coefficients of this time step
        s = reg.solver(name, cfg).set_matrix(case)         # constructor of this step: lookup-or-create by field name
        x, perf = s.solve(b, np.zeros_like(b))
        assert 1 <= perf.n_iterations < 400 and perf.final_residual < 1e-6, (name, step, perf.n_iterations, perf.final_residual)


LEDGER_EXACT = ("device_bytes", "device_blocks", "pinned_bytes", "pinned_blocks", "streams", "events", "graph_execs")


def main(steps=300, mark_every=50, size=24):
    reg = capi.Registry()
    fields = fields_of(size)
    marks = []
    for step in range(steps + 1):
        one_step(reg, fields, step)
        if step % mark_every == 0:
            free, total = reg.mem_info()
            m = {"step": step, "driver_in_use": total - free,
                 "rss_kb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, **capi.memory_ledger().as_dict()}
            marks.append(m)
            print(json.dumps(m), flush=True)
    reg.close()
    after = capi.memory_ledger().as_dict()
    # the first mark (step 0) is before the steady state: the second solve of a field may still allocate (cached
    # preconditioner, graph); drift is measured from the second mark on
    steady = marks[1:]
    drift = {k: steady[-1][k] - steady[0][k] for k in LEDGER_EXACT}
    xs = np.array([m["step"] for m in steady], dtype=float)
    ys = np.array([m["driver_in_use"] for m in steady], dtype=float)
    slope = float(np.polyfit(xs, ys, 1)[0]) if len(xs) >= 2 else 0.0
    summary = {"summary": True, "steps": steps, "fields": len(fields), "solves": (steps + 1) * len(fields),
               "ledger_drift": drift, "ledger_after_close": {k: after[k] for k in LEDGER_EXACT + ("unknown_frees",)},
               "alloc_calls_in_steady_state": {k: steady[-1][k] - steady[0][k]
                                               for k in ("device_alloc_calls", "pinned_alloc_calls", "events_created",
                                                         "graph_execs_created")},
               "driver_in_use_first": int(ys[0]), "driver_in_use_last": int(ys[-1]),
               "driver_in_use_min": int(ys.min()), "driver_in_use_max": int(ys.max()),
               "driver_slope_bytes_per_step": slope,
               "rss_growth_kb": steady[-1]["rss_kb"] - steady[0]["rss_kb"]}
    print(json.dumps(summary), flush=True)
    return 0


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(main(*a))
