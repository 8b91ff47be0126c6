#!/bin/bash
# half storage of the symmetric matrix: tests, then benches against full storage
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for E in 216 368 128 100 64; do
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --edge $E > gpurun_out/r02x_e${E}_half_$i.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --edge $E --full-storage > gpurun_out/r02x_e${E}_full_$i.json 2>/dev/null
done; done
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --precond none > gpurun_out/r02x_e216_none_half.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --precond none --full-storage > gpurun_out/r02x_e216_none_full.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02x_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    r=d["roofline"]
    print("%-22s %8.1f it/s layout=%-4s spmv %6.1f us frac %.3f moved_frac %.3f" % (f.split("/")[-1][5:-5], d["value"], r["layout"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"]))
PY
