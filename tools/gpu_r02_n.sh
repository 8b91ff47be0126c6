#!/bin/bash
# CSR-stream variants on the Voronoi (polyhedral) proxy in the library's RCM numbering + policy check
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc
timeout 900 python -m pytest tests/test_gpu_renumber.py tests/test_gpu_sell.py -m gpu -q -x 2>&1 | tail -3
for SZ in 1000000 3000000; do
  python tools/dump_pattern.py voronoi $SZ /tmp/cc/vor_$SZ.bin
  timeout 600 tools/bin/csr_tune /tmp/cc/vor_$SZ.bin 50 2>&1 | tee gpurun_out/r02n_csr_tune_vor_$SZ.txt
done
python tools/dump_pattern.py poisson 160 /tmp/cc/box160.bin 0
timeout 600 tools/bin/csr_tune /tmp/cc/box160.bin 50 2>&1 | tee gpurun_out/r02n_csr_tune_box160.txt
for SZ in 1000000 3000000; do
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --voronoi $SZ > gpurun_out/r02n_vor_$SZ.json 2> gpurun_out/r02n_vor_$SZ.err || tail -3 gpurun_out/r02n_vor_$SZ.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02n_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-24s %8.1f it/s layout=%-4s renumbered=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f sectors %s nnz %d set_matrix %.1f s" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["gather_sectors_per_entry"], c["nnz_per_gpu"], d["boundary"]["first_set_matrix_s"]))
PY
