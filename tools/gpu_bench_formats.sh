#!/bin/bash
mkdir -p gpurun_out
for F in Csr Ell; do
  python bench.py --steps 5 --warmup 1 --cpu-iters 0 --format $F > gpurun_out/bench_$F.json 2>gpurun_out/bench_$F.err
  python - "$F" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/bench_{sys.argv[1]}.json"))
r=d["roofline"]; print(sys.argv[1], "iters/s=%.1f ms/iter=%.4f spmv_ms=%.4f achieved=%.0f GB/s frac=%.3f"%(d["value"], d["cg_iteration"]["ms"], r["avg_kernel_ms"], r["achieved"], r["frac"]))
PY
done
