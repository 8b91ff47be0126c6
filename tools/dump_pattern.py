"""Development helper: writes the device-side CSR pattern of a synthetic case (after the library's
renumbering policy) for tools/csr_tune.hip.   python tools/dump_pattern.py voronoi 1000000 out.bin"""
import sys
import numpy as np
sys.path.insert(0, ".")
from ogl_amd import capi, synthetic  # noqa: E402

kind, size, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
mode = int(sys.argv[4]) if len(sys.argv) > 4 else capi.RENUMBER_AUTO
case = synthetic.voronoi_case(size) if kind == "voronoi" else synthetic.poisson_case(size)
d, loc, _, _, (ren, _) = capi.host_pattern_renumbered(case, mode)
rp = np.concatenate([[0], np.cumsum(np.bincount(loc[0], minlength=d.n_rows))]).astype(np.int32)
with open(out, "wb") as f:
    np.array([d.n_rows, d.local_nnz], dtype=np.int32).tofile(f)
    rp.tofile(f)
    loc[1].astype(np.int32).tofile(f)
print(kind, size, "rows", d.n_rows, "nnz", d.local_nnz, "renumbered", ren)
