#!/bin/bash
# octree (hex-dominant) proxy: tests, then benches against CSR-stream
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sell.py -m gpu -q -x 2>&1 | tail -3
run() { name=$1; shift; python bench.py --steps 3 --warmup 1 --cpu-iters 0 "$@" > gpurun_out/r02s_$name.json 2> gpurun_out/r02s_$name.err || tail -3 gpurun_out/r02s_$name.err; }
run oct15 --octree 1.5
run oct15_nocompress --octree 1.5 --no-compress
run oct4 --octree 4
run oct4_nocompress --octree 4 --no-compress
run oct15_append --octree 1.5 --octree-append
run oct15_append_nocompress --octree 1.5 --octree-append --no-compress
run oct4_append --octree 4 --octree-append
run oct4_append_off --octree 4 --octree-append --renumber off
run oct4_shuffle --octree 4 --shuffle 65536
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02s_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-26s %8.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f tuned %s rows %d nnz %d set_matrix %.1f s" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c["rows_sorted_by_length"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["layout_tuned_us"], c["rows_per_gpu"], c["nnz_per_gpu"], d["boundary"]["first_set_matrix_s"]))
PY
