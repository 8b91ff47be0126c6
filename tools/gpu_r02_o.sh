#!/bin/bash
# lane-masked padding + rows of a wavefront sorted by length: tests, then A/B against the base library
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc
timeout 1500 python -m pytest tests/test_gpu_sell.py tests/test_gpu_renumber.py tests/test_gpu_random_systems.py tests/test_gpu_spmv.py -m gpu -q -x 2>&1 | tail -5
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --voronoi 3000000 2>&1 | tee gpurun_out/r02o_ab_voronoi.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 2>&1 | tee gpurun_out/r02o_ab_default.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --shuffle 65536 2>&1 | tee gpurun_out/r02o_ab_shuffle.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --long-rows 0.03 2>&1 | tee gpurun_out/r02o_ab_long.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --drop-faces 0.3 2>&1 | tee gpurun_out/r02o_ab_drop.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --voronoi 1000000 2>&1 | tee gpurun_out/r02o_ab_voronoi1m.txt
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --voronoi 3000000 > gpurun_out/r02o_vor3m.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --drop-faces 0.3 > gpurun_out/r02o_drop.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 > gpurun_out/r02o_long.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02o_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-16s %8.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f nnz %d set_matrix %.1f s" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c["rows_sorted_by_length"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["nnz_per_gpu"], d["boundary"]["first_set_matrix_s"]))
PY
