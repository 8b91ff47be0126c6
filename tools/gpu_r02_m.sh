#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sell.py tests/test_gpu_random_systems.py tests/test_gpu_renumber.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 > gpurun_out/r02m_long_$i.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 --no-compress > gpurun_out/r02m_long_nocompress_$i.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/r02m_default_$i.json 2> /dev/null
done
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --long-rows 0.03 --shuffle 65536 > gpurun_out/r02m_long_shuffle.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02m_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-30s %7.1f it/s layout=%-4s renumbered=%-5s spilled=%8d spmv %6.1f us frac %.3f moved_frac %.3f" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c["spilled_entries"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"]))
PY
