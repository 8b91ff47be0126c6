#!/usr/bin/env python3
"""Measured deviation of the GPU residual history from the oracle run in the REFERENCE executor's
sequential summation order (the number behind DESIGN.md §2's tolerance schedule).

For every frozen case of tests/golden/krylov_histories.json and for the first 50 turns of the
216^3 system (BASELINE.json configs[1]) it prints, per band of the residual, the maximum relative
deviation |h_gpu - h_seq| / h_seq, and the iteration counts.  Run on the GPU box:

  python tools/parity_deviation.py > gpurun_out/r02_parity_deviation.txt
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_golden  # noqa: E402
from ogl_amd import capi, synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import oracle_csr  # noqa: E402


def bands(h_gpu, h_seq):
    m = min(h_gpu.size, h_seq.size)
    dev = np.abs(h_gpu[:m] - h_seq[:m]) / np.maximum(np.abs(h_seq[:m]), 1e-300)
    rel = h_seq[:m] / h_seq[0]
    out = [("first 5 checks", dev[:5].max())]
    for lo, hi in [(1e-1, 2.0), (1e-3, 1e-1), (1e-5, 1e-3), (1e-7, 1e-5), (0.0, 1e-7)]:
        sel = (rel > lo) & (rel <= hi)
        if sel.any():
            out.append((f"residual in ({lo:g}, {hi:g}] of its start ({int(sel.sum())} checks)", dev[sel].max()))
    return out, dev


def gpu_solve(reg, name, case, b, solver, precond, extra, **kw):
    cfg = capi.default_config(
        solver={"cg": capi.SOLVER_CG, "bicgstab": capi.SOLVER_BICGSTAB, "gmres": capi.SOLVER_GMRES}[solver],
        preconditioner=capi.PRECOND_BJ if precond == "bj" else capi.PRECOND_NONE,
        krylov_dim=extra.get("krylov_dim", 0), export_res=1, adapt_min_iter=0, matrix_format=capi.FORMAT_CSR,
        **kw)
    s = reg.solver(name, cfg).set_matrix(case)
    x, perf = s.solve(b, np.zeros_like(b))
    return x, perf, s.history()


def main():
    orc.build()
    orc.set_reduction(orc.REDUCE_SEQUENTIAL)
    reg = capi.Registry()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "krylov_histories.json")))["cases"]
    print("# GPU residual history vs the oracle in sequential (reference-executor) order")
    print("# max relative deviation per band of the residual; iteration counts gpu / sequential oracle")
    for name, edge, sym, solver, precond, extra in gen_golden.KRYLOV_CASES:
        case = synthetic.poisson_case(edge, symmetric=sym)
        rp, cols, vals = oracle_csr(orc, case)
        b = orc.spmv(rp, cols, vals, gen_golden.x_dyadic(case.n_cells))
        x, perf, hist = gpu_solve(reg, name, case, b, solver, precond, extra, **gen_golden.SOLVE_KW)
        h_seq = np.array([float.fromhex(v) for v in gold[name]["history"]])
        n_seq = gold[name]["n_iterations"]
        n_gpu = perf.n_iterations * (2 if solver == "bicgstab" else 1)
        print(f"\n{name}: {case.n_cells} rows, checks gpu {hist.size} / oracle {h_seq.size} "
              f"(reported iterations {n_gpu} / {n_seq})")
        for label, v in bands(hist, h_seq)[0]:
            print(f"   {label:60s} {v:.3e}")
    # ---- BASELINE size: first 50 turns, CG + BJ
    edge = int(os.environ.get("OGL_DEV_EDGE", "216"))
    case = synthetic.poisson_case(edge)
    b, xs = synthetic.rhs_for_x_star(case)
    kw = dict(tolerance=0.0, rel_tol=0.0, max_iter=50)
    for precond in ("none", "bj"):
        x, perf, hist = gpu_solve(reg, f"big_{precond}", case, b, "cg", precond, {}, **kw)
        rows, cols, perm = orc.init_local_sparsity(case.n_cells, case.upper_addr, case.lower_addr, True)
        vals = orc.update_local_matrix_data(case.diag, case.upper, None, [], perm)
        rp = orc.rowptr_from_rows(case.n_cells, rows)
        A = orc.DistMatrix(rp, cols, vals)
        inv = orc.jacobi_generate_scalar(rp, cols, vals) if precond == "bj" else None
        ref = orc.cg(A, b, np.zeros_like(b), inv, **kw)
        out, dev = bands(hist, ref.history)
        print(f"\ncg_{precond}_{edge} (BASELINE configs[1] size, {case.n_cells} rows), first 50 turns: "
              f"history[50] gpu {hist[-1]:.6e} oracle {ref.history[-1]:.6e}")
        for label, v in out:
            print(f"   {label:60s} {v:.3e}")
        print("   per check:", " ".join(f"{v:.1e}" for v in dev))
        print(f"   max |x_gpu - x_oracle| = {np.abs(x - ref.x).max():.3e}")
        arbiter(f"cg_{precond}_{edge}, 50 turns", hist, ref.history, A, b, inv, kw)
    # ---- 64^3 to convergence under the arbiter
    case = synthetic.poisson_case(64)
    b, xs = synthetic.rhs_for_x_star(case)
    kw = dict(tolerance=1e-9, rel_tol=0.0, max_iter=2000)
    x, perf, hist = gpu_solve(reg, "arb64", case, b, "cg", "bj", {}, **kw)
    rp, cols, vals = oracle_csr(orc, case)
    A = orc.DistMatrix(rp, cols, vals)
    inv = orc.jacobi_generate_scalar(rp, cols, vals)
    ref = orc.cg(A, b, np.zeros_like(b), inv, **kw)
    arbiter("cg_bj_64 to 1e-9", hist, ref.history, A, b, inv, kw)
    reg.close()


def arbiter(label, h_gpu, h_seq, A, b, inv, kw):
    """Both orders against the oracle's EXACT mode (error-free transformations, rounded once): which one is closer to the
    exact history (tests/test_gpu_exact_arbiter.py asserts what is printed here)."""
    orc.set_reduction(orc.REDUCE_EXACT)
    try:
        ex = orc.cg(A, b, np.zeros_like(b), inv, **kw).history
    finally:
        orc.set_reduction(orc.REDUCE_SEQUENTIAL)
    m = min(ex.size, h_gpu.size, h_seq.size)
    d_gpu = np.abs(h_gpu[:m] - ex[:m]) / ex[:m]
    d_seq = np.abs(h_seq[:m] - ex[:m]) / ex[:m]
    print(f"\n== arbiter, {label}: |history - exact| / exact per check ({m} checks; exact = TwoSum / TwoProduct accumulation)")
    print("   check  residual/start   sequential (reference executor's order)   device tree")
    step = max(1, m // 25)
    for k in list(range(0, m, step)) + [m - 1]:
        print(f"   {k:5d}  {ex[k] / ex[0]:.3e}      {d_seq[k]:.3e}                                {d_gpu[k]:.3e}")
    early = ex[:m] / ex[0] > 1e-3
    print(f"   running maxima: residual above 1e-3 of its start ({int(early.sum())} checks): sequential {d_seq[early].max():.3e}, "
          f"device {d_gpu[early].max():.3e}; all checks: sequential {d_seq.max():.3e}, device {d_gpu.max():.3e}")


if __name__ == "__main__":
    main()
