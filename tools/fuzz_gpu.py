#!/usr/bin/env python3
"""Randomised sweep of the GPU path against the oracle (run by hand on a GPU box; the committed tests are the fixed
cases): random patterns (boxes, multi-block meshes, random bands, scattered extra faces; odd and even row counts),
random solver / preconditioner / layout switches / turn shapes.  Every case: SpMV bit-identical to the oracle's, the
residual history and x of a short solve bit-identical to the oracle run in the device's reduction tree.

  python tools/fuzz_gpu.py [cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ogl_amd import capi, synthetic          # noqa: E402
from oracle import oracle as orc              # noqa: E402
from helpers import (blocked, oracle_csr, oracle_matrix, oracle_matrix_renumbered,   # noqa: E402
                     oracle_precond_renumbered, to_new)


def extra_faces(case, count, rng):
    have = set(zip(case.lower_addr.tolist(), case.upper_addr.tolist()))
    lo, up = [], []
    n = case.n_cells
    tries = 0
    while len(lo) < count and n > 2 and tries < 20 * count:   # (a tiny mesh has fewer free pairs than `count`)
        tries += 1
        a, b = sorted(int(v) for v in rng.integers(0, n, 2))
        if a != b and (a, b) not in have:
            have.add((a, b)); lo.append(a); up.append(b)
    if not lo:
        return case
    la = np.concatenate([case.lower_addr, np.array(lo, np.int32)])
    ua = np.concatenate([case.upper_addr, np.array(up, np.int32)])
    coef = np.concatenate([case.upper, rng.uniform(-1.0, -0.25, len(lo))])
    low = None if case.lower is None else np.concatenate([case.lower, rng.uniform(-1.0, -0.25, len(lo))])
    order = np.lexsort((ua, la))
    diag = case.diag.copy()
    np.add.at(diag, np.array(lo), 1.1); np.add.at(diag, np.array(up), 1.1)
    return synthetic.LduCase(n, la[order].astype(np.int32), ua[order].astype(np.int32), diag, coef[order],
                             None if low is None else low[order], [], case.global_index, case.global_n)


SCALE = int(os.environ.get("OGL_FUZZ_SCALE", "1"))   # 2-3: systems of several hundred chunks


def random_case(rng):
    kind = rng.choice(["box", "blocks", "band", "box+faces", "blocks+faces", "line", "periodic"])
    sym = bool(rng.integers(0, 4) != 0)
    if kind.startswith("box"):
        g = [int(rng.integers(1, 40 * SCALE)) for _ in range(3)]
        c = synthetic.poisson_block(*g, symmetric=sym, **({} if sym else dict(off_upper=-0.9, off_lower=-1.1)))
    elif kind.startswith("blocks"):
        nb = int(rng.integers(2, 4))
        c = synthetic.multi_block_case([int(rng.integers(2, 40 * SCALE)) for _ in range(nb)], int(rng.integers(1, 20 * SCALE)),
                                      int(rng.integers(1, 20 * SCALE)))
        if not sym:
            c.lower = c.upper * 1.2
    elif kind == "periodic":   # a cyclic patch pair on the same rank: its entries are merged into the local rows
        g = [int(rng.integers(2, 30 * SCALE)) for _ in range(3)]
        c = synthetic.poisson_block(*g, symmetric=sym, periodic_x=True, **({} if sym else dict(off_upper=-0.9, off_lower=-1.1)))
    elif kind == "line":
        c = synthetic.poisson_block(int(rng.integers(1, 3000 * SCALE * SCALE)), 1, 1, symmetric=sym, **({} if sym else dict(off_upper=-0.9, off_lower=-1.1)))
    else:
        c = synthetic.random_global_case(int(rng.integers(2, 3000 * SCALE)), int(rng.integers(1, 5)), int(rng.choice([2, 4, 9, 60])),
                                         symmetric=sym, seed=int(rng.integers(0, 1 << 30)))
    if kind.endswith("+faces"):
        c = extra_faces(c, int(rng.integers(1, 60 * SCALE * SCALE)), rng)
    if c.interfaces:
        return kind, c
    # random coefficients, diagonally dominant
    c.upper = rng.uniform(-1.0, -0.25, c.upper.size)
    if c.lower is not None:
        c.lower = rng.uniform(-1.0, -0.25, c.upper.size)
    deg = np.bincount(c.lower_addr, minlength=c.n_cells) + np.bincount(c.upper_addr, minlength=c.n_cells)
    c.diag = deg * 1.0 + rng.uniform(0.5, 1.5, c.n_cells)
    return kind, c


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    orc.build()
    dry = os.environ.get("OGL_FUZZ_DRY") is not None   # CPU only: the oracle's half of every case, timed (which case is slow?)
    reg = None if dry else capi.Registry()
    chunk = 512 if dry else capi.lib().ogl_reduction_chunk_rows()
    import time
    bad = 0
    for it in range(n_cases):
        kind, case = random_case(rng)
        solver = rng.choice(["cg", "bicg", "gmres"]) if case.lower is None else rng.choice(["bicg", "gmres"])
        pc = str(rng.choice(["none", "bj1", "bj1", "bjk", "isai", "gisai"]))
        if pc == "isai" and case.lower is not None:
            pc = "gisai"                                  # (ISAI is the SPD variant)
        precond = {"none": capi.PRECOND_NONE, "bj1": capi.PRECOND_BJ, "bjk": capi.PRECOND_BJ, "isai": capi.PRECOND_ISAI,
                   "gisai": capi.PRECOND_GISAI}[pc]
        block = int(rng.integers(2, 9)) if pc == "bjk" else 1
        cfgkw = dict(max_block_size=block, solver={"cg": capi.SOLVER_CG, "bicg": capi.SOLVER_BICGSTAB, "gmres": capi.SOLVER_GMRES}[solver],
                     preconditioner=precond, tolerance=1e-12, rel_tol=0.0, max_iter=int(rng.integers(1, 40)), export_res=1,
                     adapt_min_iter=0, update_init_guess=1,
                     # (round 4: the library's own renumbering in a third of the cases -- preconditioner structures stay
                     #  those of the caller's numbering -- and ISAI patterns of S^1 ... S^3, rows of up to 512 entries)
                     renumber=capi.RENUMBER_ON if rng.integers(0, 3) == 0 and not case.interfaces else capi.RENUMBER_OFF,
                     # (powers above 1 on the stencil-like kinds only: on a random band S^3 reaches hundreds of entries per
                     #  row and the ORACLE's dense solves take minutes per case)
                     sparsity_power=int(rng.integers(1, 4)) if pc in ("isai", "gisai") and kind in ("box", "blocks", "line", "periodic") else 1,
                     compress_indices=int(rng.integers(0, 4) != 0), symmetric_half=int(rng.integers(0, 4) != 0),
                     matrix_format=int(rng.choice([capi.FORMAT_CSR, capi.FORMAT_CSR, capi.FORMAT_ELL])))
        if solver == "gmres":
            cfgkw["krylov_dim"] = int(rng.integers(2, 12))
        props = {"streamAboveBytes": float(rng.choice([0.0, 1e18])), "fusedFinalizers": float(rng.integers(0, 2)),
                 "fusedTurn": float(rng.integers(0, 2)), "fusedTurnBig": float(rng.integers(0, 2)),
                 "hipGraph": float(rng.integers(0, 2)),
                 # (round 5: folded GKOBiCGStab / GKOGMRES turns, W rows sorted inside the windows, one-pass block Jacobi
                 #  through the permutation, STREAM decision on the turn's working set, band-aware workgroup order)
                 "bicgFold": float(rng.integers(0, 2)), "gmresFold": float(rng.integers(0, 2)),
                 "isaiSortRows": float(rng.integers(0, 2)), "bjFusedPerm": float(rng.integers(0, 2)),
                 "streamTurnSet": float(rng.integers(0, 2)), "spmvBandRows": float(rng.choice([-1.0, 0.0, 4096.0])),
                 # (round 6: leader finalisation -- with fusedFinMaxChunks 0 every system of 48 chunks or more takes it --
                 #  and the CSR-stream kernel's LDS rounds)
                 "fusedFinMaxChunks": float(rng.choice([1024.0, 0.0])), "leadFinalizers": float(rng.integers(0, 2)),
                 "leadEarlyLoads": float(rng.integers(0, 2)), "gmresLead": float(rng.integers(0, 2)),
                 "spmvLdsRounds": float(rng.integers(1, 3)),
                 # (x advanced by every K-th head from a ring of K search directions, or every turn)
                 "deferX": float(rng.choice([0.0, 2.0, 4.0, 8.0]))}
        tag = f"case {it}: {kind} n={case.n_cells} sym={case.lower is None} {solver} precond={pc}/{block} " \
              f"{ {k: cfgkw[k] for k in ('compress_indices', 'symmetric_half', 'matrix_format', 'max_iter', 'renumber', 'sparsity_power')} } {props}"
        only = os.environ.get("OGL_FUZZ_ONLY")
        if only is not None and it != int(only):
            for _ in range(2):
                rng.uniform(-1, 1, case.n_cells)      # (keep the random stream of the skipped case's x and b)
            continue
        try:
            t_case = time.time()
            if dry and os.environ.get("OGL_FUZZ_DRY") == "2":
                print("start", tag, flush=True)
            if not dry:
                s = reg.solver(f"f{it}", capi.default_config(**cfgkw))
                for k, v in props.items():
                    s.set_property(k, v)
                s.set_matrix(case)
            x = rng.uniform(-1, 1, case.n_cells)
            b = rng.uniform(-1, 1, case.n_cells)
            y = None if dry else s.spmv(x)
            new_id = None if dry else s.renumbering()
            if new_id is None:
                rp, cols, vals = oracle_csr(orc, case)
                A, _ = oracle_matrix(orc, case)
                assert dry or np.array_equal(y, orc.spmv(rp, cols, vals, x)), "spmv differs"
                mk = lambda *a, **k_: orc.Precond(rp, cols, vals, *a, **k_)       # noqa: E731
                b_o, x_o, back = b, x.copy(), (lambda v: v)
            else:   # the oracle on the system in the numbering the library reports, vectors in that numbering
                A, (rp, cols, vals) = oracle_matrix_renumbered(orc, case, new_id)
                assert np.array_equal(y, orc.spmv(rp, cols, vals, to_new(x, new_id))[new_id]), "spmv differs (renumbered)"
                mk = lambda *a, **k_: oracle_precond_renumbered(orc, case, rp, cols, vals, new_id, *a, **k_)   # noqa: E731
                b_o, x_o, back = to_new(b, new_id), to_new(x, new_id), (lambda v: v[new_id])
            kw = dict(tolerance=1e-12, rel_tol=0.0, max_iter=cfgkw["max_iter"])
            too_wide = False
            if pc == "none":
                P = None
            elif pc == "bj1":
                P = mk(1) if solver == "gmres" else orc.jacobi_generate_scalar(rp, cols, vals)
            elif pc == "bjk":
                P = mk(block)
            else:
                try:
                    P = mk(isai="spd" if pc == "isai" else "general", sparsity_power=cfgkw["sparsity_power"])
                except ValueError:
                    too_wide = True       # a row of W wider than 2048: both sides refuse
            if too_wide and not dry:
                try:
                    s.solve(b, x.copy())
                    raise AssertionError("a W row wider than 2048 was accepted")
                except capi.OglError as e:
                    assert e.status == capi.ERR_UNSUPPORTED, e
                continue
            with blocked(orc, chunk):
                if solver == "cg":
                    ref = orc.cg(A, b_o, x_o, P, **kw)
                elif solver == "bicg":
                    ref = orc.bicgstab(A, b_o, x_o, P, **kw)
                else:
                    ref = orc.gmres(A, b_o, x_o, P, krylov_dim=cfgkw["krylov_dim"], **kw)
            ref.x = back(ref.x)
            if dry:
                if time.time() - t_case > 2.0:
                    print(f"slow ({time.time() - t_case:.1f} s):", tag, flush=True)
                continue
            xs, perf = s.solve(b, x.copy())
            if solver != "bicg":   # (GKOBiCGStab reports half of its checks, as the reference does)
                assert perf.n_iterations == ref.n_iterations, f"iterations {perf.n_iterations} vs {ref.n_iterations}"
            if only is not None:
                print("gpu history", s.history(), "\noracle history", ref.history, "\ngpu x", xs, "oracle x", ref.x)
            assert np.array_equal(s.history(), ref.history, equal_nan=True), "history differs"
            assert np.array_equal(xs, ref.x, equal_nan=True), "x differs"
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("FAIL", tag, "->", repr(e)[:300], flush=True)
    print(f"{n_cases - bad} / {n_cases} cases bit-identical (seed {seed})")
    if reg is not None:
        reg.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
