#!/bin/bash
# band-aware XCD order of the compressed SpMV: A/B through OGL_NO_BAND_ORDER, then PMC
mkdir -p gpurun_out
rm -f gpurun_out/r02u_*.json
for E in 216 368 128 100 160; do
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --edge $E > gpurun_out/r02u_e${E}_band_$i.json 2>/dev/null
OGL_NO_BAND_ORDER=1 python bench.py --steps 3 --warmup 1 --cpu-iters 0 --edge $E > gpurun_out/r02u_e${E}_noband_$i.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02u_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    r=d["roofline"]
    print("%-22s %8.1f it/s spmv %6.1f us frac %.3f" % (f.split("/")[-1][5:-5], d["value"], 1e3*r["avg_kernel_ms"], r["frac"]))
PY
bash tools/gpu_r02_v.sh
