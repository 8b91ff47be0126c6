#!/bin/bash
mkdir -p gpurun_out
for W in "--octree 1.5" "--octree 4" "--long-rows 0.03"; do
echo "== $W (A: base = spill by entries only; in-tree: spill has a fixed cost per chunk)"
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 $W 2>&1
echo "-- renumber off"
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 $W --renumber off 2>&1
done | tee gpurun_out/r02t_ab.txt
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --octree 1.5 > gpurun_out/r02t_oct15.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --octree 1.5 --renumber off > gpurun_out/r02t_oct15_off.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 > gpurun_out/r02t_long.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 --renumber off > gpurun_out/r02t_long_off.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02t_*.json")):
    d=json.load(open(f)); r=d["roofline"]; c=d["config"]
    print("%-16s %8.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f tuned %s" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c["rows_sorted_by_length"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["layout_tuned_us"]))
PY
