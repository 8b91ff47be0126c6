import time, numpy as np, sys
sys.path.insert(0, '.')
from ogl_amd import capi, synthetic
case = synthetic.voronoi_case(200000)
reg = capi.Registry()
b = np.ones(case.n_cells)
for power in (1, 2):
    cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_GISAI, sparsity_power=power, tolerance=1e-8, rel_tol=0.0, max_iter=300, export_res=0)
    s = reg.solver(f"p{power}", cfg)
    t0 = time.time(); s.set_matrix(case); t1 = time.time()
    try:
        x, perf = s.solve(b, np.zeros_like(b)); t2 = time.time()
        x, perf2 = s.solve(b, np.zeros_like(b)); t3 = time.time()
        print(f"GISAI sparsityPower {power}: set_matrix {t1-t0:.2f} s, first solve {t2-t1:.2f} s, second solve {t3-t2:.2f} s, iterations {perf.n_iterations}, "
              f"wide rows {s.get_property('isaiWideRows'):.0f}, huge rows {s.get_property('isaiHugeRows'):.0f}, solve_ms {perf2.t_solve_ms:.1f}")
    except capi.OglError as e:
        print(f"GISAI sparsityPower {power}: refused: {e}")
