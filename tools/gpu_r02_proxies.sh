#!/bin/bash
# the unstructured / mixed proxies at HEAD (end of round 2)
mkdir -p gpurun_out /tmp/cc; rm -f gpurun_out/r02pr_*.json
export OGL_CASE_CACHE_DIR=/tmp/cc
run() { name=$1; shift; python bench.py --steps 3 --warmup 1 --cpu-iters 0 "$@" > gpurun_out/r02pr_$name.json 2> gpurun_out/r02pr_$name.err || tail -3 gpurun_out/r02pr_$name.err; }
run shuffle65536 --shuffle 65536
run shuffle65536_nocompress --shuffle 65536 --no-compress
run drop --drop-faces 0.3
run drop_nocompress --drop-faces 0.3 --no-compress
run long --long-rows 0.03
run long_nocompress --long-rows 0.03 --no-compress
run long_shuffle --long-rows 0.03 --shuffle 65536
run oct15 --octree 1.5
run oct15_nocompress --octree 1.5 --no-compress
run oct4 --octree 4
run oct4_append --octree 4 --octree-append
run vor1m --voronoi 1000000
run vor3m --voronoi 3000000
run vor3m_off --voronoi 3000000 --renumber off
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02pr_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-26s %8.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f tuned %s" % (
        f.split("/")[-1][6:-5], d["value"], r["layout"], c["renumbered"], c["rows_sorted_by_length"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"],
        None if not c["layout_tuned_us"] else {k: round(v,1) for k,v in c["layout_tuned_us"].items()}))
PY
