export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_systems.py tests/test_gpu_renumber.py -m gpu -q -x -k "jacobi or random or renumber or caching" 2>&1 | tail -4
for CN in 1 0 1 0; do
python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size 4 --prop precondCallerNumbering=$CN > gpurun_out/r04j_bj4_cn$CN.json 2> gpurun_out/r04j_bj4_cn$CN.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04j_bj4_cn$CN.json")); print("cg+BJ4 128^3 shuffled, precondCallerNumbering=$CN", d["value"], "turns/s, turn us", 1e3*d["solver_turn"]["ms"])
PY
done 2>&1 | tee gpurun_out/r04j_precond_numbering.txt
python - <<'PY'
import numpy as np
from ogl_amd import capi, synthetic
case = synthetic.renumber_case(synthetic.poisson_case(128, symmetric=False), 65536)
reg = capi.Registry()
for cn in (1.0, 0.0):
    cfg = capi.default_config(solver=capi.SOLVER_BICGSTAB, preconditioner=capi.PRECOND_ISAI, tolerance=0.0, rel_tol=0.0, max_iter=5)
    s = reg.solver(f"u{cn}", cfg); s.set_property("precondCallerNumbering", cn); s.set_matrix(case)
    b = np.ones(case.n_cells); s.solve(b, np.zeros_like(b))
    print("precondCallerNumbering", cn, "W compressed", s.get_property("isaiWCompressed"), "Wt compressed", s.get_property("isaiWtCompressed"))
PY
