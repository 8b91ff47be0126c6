#!/bin/bash
mkdir -p gpurun_out
for E in 32 64 100 128 160; do
  python bench.py --steps 5 --warmup 1 --cpu-iters 0 --edge $E --iters 200 > gpurun_out/bench_e$E.json 2>/dev/null
  python - "$E" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/bench_e{sys.argv[1]}.json"))
r=d["roofline"]; print("edge",sys.argv[1],"rows",d["config"]["rows_per_gpu"],"iters/s=%.0f us/iter=%.1f spmv_us=%.1f cg_frac_of_peak=%.3f"%(d["value"], 1e3*d["cg_iteration"]["ms"], 1e3*r["avg_kernel_ms"], d["cg_iteration"]["frac_of_peak"]))
PY
done
