import numpy as np, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from ogl_amd import capi, synthetic
from oracle import oracle as orc
from helpers import oracle_csr
n = 2048
lower = [r for r in range(n - 1)] + [r for r in range(7, n, 512) for _ in range(2, 200) if r + 199 < n]
upper = [r + 1 for r in range(n - 1)] + [r + d for r in range(7, n, 512) for d in range(2, 200) if r + 199 < n]
order = np.lexsort((upper, lower))
lower, upper = np.array(lower, np.int32)[order], np.array(upper, np.int32)[order]
rng = np.random.default_rng(3)
case = synthetic.LduCase(n, lower, upper, rng.uniform(300, 400, n), rng.uniform(-1, 1, len(lower)), None)
reg = capi.Registry()
cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, matrix_format=capi.FORMAT_CSR, renumber=0)
s = reg.solver("d", cfg).set_matrix(case)
rp, cols, vals = oracle_csr(orc, case)
x = rng.uniform(-1, 1, n)
y = s.spmv(x); ref = orc.spmv(rp, cols, vals, x)
bad = np.flatnonzero(y != ref)
print("bad rows", bad, "layout", s.get_property("spmvLayout"), "spilled", s.get_property("sellSpilledEntries"))
for r in bad:
    c = cols[rp[r]:rp[r+1]]; v = vals[rp[r]:rp[r+1]]
    part = np.cumsum(v * x[c])
    k = np.argmin(np.abs(part - y[r]))
    print("row", r, "len", c.size, "y", y[r], "ref", ref[r], "closest prefix", k + 1, part[k], "diff", y[r]-ref[r])
    # which single entries could explain the difference?
    d = ref[r] - y[r]
    j = np.argmin(np.abs(v * x[c] - d)); print("   single missing entry candidate idx", j, v[j]*x[c[j]], "col", c[j])
