#!/bin/bash
# lane-masked padding + rows of a wavefront sorted by length + one-off layout timing: tests, then benches
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc
timeout 2400 python -m pytest tests/test_gpu_sell.py tests/test_gpu_renumber.py tests/test_gpu_random_systems.py tests/test_gpu_fullsize_configs.py -m gpu -q -x 2>&1 | tail -5
run() { name=$1; shift; python bench.py --steps 3 --warmup 1 --cpu-iters 0 "$@" > gpurun_out/r02p_$name.json 2> gpurun_out/r02p_$name.err || tail -3 gpurun_out/r02p_$name.err; }
run vor3m --voronoi 3000000
run vor3m_force --voronoi 3000000 --force-compress
run vor1m --voronoi 1000000
run drop --drop-faces 0.3
run long --long-rows 0.03
run long_shuffle --long-rows 0.03 --shuffle 65536
run long_shuffle_nocompress --long-rows 0.03 --shuffle 65536 --no-compress
run shuffle --shuffle 65536
run default
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02p_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-24s %8.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f tuned %s set_matrix %.1f s" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c["rows_sorted_by_length"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["layout_tuned_us"], d["boundary"]["first_set_matrix_s"]))
PY
