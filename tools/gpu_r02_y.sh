#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r02y_*.json
timeout 900 python -m pytest tests/test_gpu_sym.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
for E in 216 368 160 128 100; do
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --edge $E > gpurun_out/r02y_e${E}_band_$i.json 2>/dev/null
OGL_NO_BAND_ORDER=1 python bench.py --steps 3 --warmup 1 --cpu-iters 0 --edge $E > gpurun_out/r02y_e${E}_noband_$i.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02y_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    r=d["roofline"]
    print("%-22s %8.1f it/s layout=%-4s spmv %6.1f us frac %.3f moved_frac %.3f" % (f.split("/")[-1][5:-5], d["value"], r["layout"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"]))
PY
