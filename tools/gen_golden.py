#!/usr/bin/env python3
"""Generates the self-made golden fixtures.

  tests/golden/krylov_histories.json   frozen residual histories of the ORACLE run in the reference
                                       executor's sequential order: GKOCG none / BJ at 16^3, 32^3,
                                       64^3, GKOBiCGStab (asym) 16^3, GKOGMRES(30) + BJ 16^3.
  tests/golden/host_matrix_generated.json   LDU -> row-major pattern / ldu_mapping / coefficients of
                                       3^3, 4^3, 5x4x3 boxes (sym + asym) and one case with a cyclic
                                       patch pair.

PROVENANCE (stated in both files): these are the ORACLE's outputs, frozen.  Nothing in
/root/reference holds Krylov numbers (test/validation.json:10-42 asserts completion only) and the
reference's free functions cannot be built here without stand-ins for the Ginkgo / OpenFOAM headers,
so the fixtures do not lift the parity grade; what they do is make an accidental change of BOTH the
oracle and the kernels visible, and the host-matrix ones are cross-checked by the product's
independently written algorithm (tests/test_golden_generated.py).

Inputs are machine independent: the exact solution is the dyadic-rational vector
x*_i = ((i * 7919) mod 1024) / 1024 - 1/2 and b = A x* is formed by the oracle's own row loop, so
no libm call is involved.  Doubles are stored as C99 hex floats (bit exact).

  python tools/gen_golden.py            # rewrites both files
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ogl_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import oracle_csr, orc_ifaces  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def x_dyadic(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 7919) % 1024).astype(np.float64) / 1024.0 - 0.5


def hexes(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64)]


KRYLOV_CASES = [
    # name, edge, symmetric, solver, preconditioner, kwargs
    ("cg_none_16", 16, True, "cg", "none", {}),
    ("cg_bj_16", 16, True, "cg", "bj", {}),
    ("cg_none_32", 32, True, "cg", "none", {}),
    ("cg_bj_32", 32, True, "cg", "bj", {}),
    ("cg_none_64", 64, True, "cg", "none", {}),
    ("cg_bj_64", 64, True, "cg", "bj", {}),
    ("bicgstab_bj_16_asym", 16, False, "bicgstab", "bj", {}),
    ("gmres30_bj_16", 16, True, "gmres", "bj", {"krylov_dim": 30}),
]
SOLVE_KW = dict(tolerance=1e-10, rel_tol=0.0, max_iter=1000)


def krylov_case(edge, sym, solver, precond, extra):
    """Shared by the generator and tests/test_golden_generated.py."""
    case = synthetic.poisson_case(edge, symmetric=sym)
    rp, cols, vals = oracle_csr(orc, case)
    A = orc.DistMatrix(rp, cols, vals)
    xs = x_dyadic(case.n_cells)
    b = orc.spmv(rp, cols, vals, xs)
    x0 = np.zeros_like(b)
    if solver == "gmres":
        P = orc.Precond(rp, cols, vals, 1) if precond == "bj" else None
        res = orc.gmres(A, b, x0, P, **SOLVE_KW, **extra)
    else:
        inv = orc.jacobi_generate_scalar(rp, cols, vals) if precond == "bj" else None
        res = (orc.cg if solver == "cg" else orc.bicgstab)(A, b, x0, inv, **SOLVE_KW)
    return case, xs, b, res


def gen_krylov():
    out = {"_provenance": "ORACLE outputs, frozen by tools/gen_golden.py (sequential = reference-executor "
                          "summation order). Not reference outputs: the reference holds no Krylov numbers "
                          "(test/validation.json:10-42); parity of the Krylov arithmetic stays unpinned.",
           "_inputs": "poisson_case(edge) of ogl_amd/synthetic.py (BASELINE.md section 3 matrix); "
                      "x*_i = ((i*7919) mod 1024)/1024 - 1/2; b = A x* by the oracle's row loop; x0 = 0; "
                      "tolerance 1e-10, relTol 0, maxIter 1000, evalFrequency 1",
           "cases": {}}
    orc.set_reduction(orc.REDUCE_SEQUENTIAL)
    for name, edge, sym, solver, precond, extra in KRYLOV_CASES:
        case, xs, b, res = krylov_case(edge, sym, solver, precond, extra)
        err = float(np.abs(res.x - xs).max())
        out["cases"][name] = {
            "edge": edge, "symmetric": sym, "solver": solver, "preconditioner": precond, **extra,
            "n_iterations": res.n_iterations, "norm_factor": float(res.norm_factor).hex(),
            "initial_residual": float(res.initial_residual).hex(),
            "final_residual": float(res.final_residual).hex(),
            "history": hexes(res.history),
            "x_checksum": float(np.sum(res.x)).hex(),      # numpy pairwise sum of the final iterate
            "x_probe": hexes(res.x[:: max(1, case.n_cells // 16)][:16]),
            "max_error_vs_x_star": err,
        }
        print(f"{name}: {res.n_iterations} checks, final {res.final_residual:.3e}, |x - x*|max {err:.2e}")
    with open(os.path.join(GOLDEN, "krylov_histories.json"), "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")


HOST_CASES = [
    ("box3_sym", dict(gx=3, gy=3, gz=3), True),
    ("box3_asym", dict(gx=3, gy=3, gz=3, off_upper=-0.9, off_lower=-1.1), False),
    ("box4_sym", dict(gx=4, gy=4, gz=4), True),
    ("box4_asym", dict(gx=4, gy=4, gz=4, off_upper=-0.9, off_lower=-1.1), False),
    ("box5x4x3_sym", dict(gx=5, gy=4, gz=3), True),
    ("box5x4x3_asym", dict(gx=5, gy=4, gz=3, off_upper=-0.9, off_lower=-1.1), False),
    ("box4x3x2_cyclic_asym", dict(gx=4, gy=3, gz=2, periodic_x=True, off_upper=-0.9, off_lower=-1.1), False),
    ("box4x4x2_rank1of2", dict(gx=4, gy=4, gz=2, px=2, rank=1), True),     # one processor interface
]


def host_case(kw, sym):
    return synthetic.poisson_block(symmetric=sym, **kw)


def gen_host():
    out = {"_provenance": "ORACLE outputs (oracle/ogl_oracle.c restating HostMatrixFreeFunctions.C:105-201 "
                          "and HostMatrix.C:385-466,468-589,592-732), frozen by tools/gen_golden.py and "
                          "cross-checked by the product's independently written host_matrix.cpp. Not "
                          "outputs of the reference itself (it does not build here).",
           "cases": {}}
    for name, kw, sym in HOST_CASES:
        case = host_case(kw, sym)
        ifs = orc_ifaces(orc, case)
        rows, cols, perm = orc.init_local_sparsity_pattern(case.n_cells, case.upper_addr, case.lower_addr,
                                                           case.symmetric, ifs)
        vals = orc.update_local_matrix_data(case.diag, case.upper, case.lower, ifs, perm)
        nl_rows, nl_cols, nl_perm = orc.init_non_local_sparsity(ifs)
        nl_vals = orc.update_non_local_matrix_data(ifs, nl_perm)
        ids, sizes, send = orc.create_communication_pattern(ifs)
        out["cases"][name] = {
            "block": kw, "symmetric": sym,
            "lower_addr": case.lower_addr.tolist(), "upper_addr": case.upper_addr.tolist(),
            "rows": rows.tolist(), "cols": cols.tolist(), "ldu_mapping": perm.tolist(),
            "coeffs": hexes(vals),
            "non_local": {"rows": nl_rows.tolist(), "cols": nl_cols.tolist(),
                          "ldu_mapping": nl_perm.tolist(), "coeffs": hexes(nl_vals)},
            "comm": {"target_ids": ids.tolist(), "target_sizes": sizes.tolist(), "send_idxs": send.tolist()},
        }
        print(f"{name}: {case.n_cells} rows, {rows.size} local entries, {nl_rows.size} non-local")
    with open(os.path.join(GOLDEN, "host_matrix_generated.json"), "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")


if __name__ == "__main__":
    orc.build()
    gen_host()
    gen_krylov()
