#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log | cut -c1-250
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 > gpurun_out/r02l_long.json 2> gpurun_out/r02l_long.err || tail -3 gpurun_out/r02l_long.err
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 --no-compress > gpurun_out/r02l_long_nocompress.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 --shuffle 65536 > gpurun_out/r02l_long_shuffle.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/r02l_default.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02l_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-30s %7.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f nnz %d" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c.get("rows_sorted_by_length"), c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["nnz_per_gpu"]))
PY
