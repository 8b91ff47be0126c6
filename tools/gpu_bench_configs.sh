#!/bin/bash
# Single-GPU measurements of the other BASELINE.json configs' solver/preconditioner pairs
mkdir -p gpurun_out
run() {
  TAG=$1; shift
  python bench.py --steps 3 --warmup 2 --cpu-iters 0 "$@" > gpurun_out/cfg_$TAG.json 2> gpurun_out/cfg_$TAG.err || { echo "$TAG FAILED"; tail -3 gpurun_out/cfg_$TAG.err; return; }
  python - "$TAG" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/cfg_{sys.argv[1]}.json"))
r=d["roofline"]
print("%-26s turns/s=%8.1f  ms/turn=%.4f  spmv_ms=%.4f  | %s" % (sys.argv[1], d["value"], d["cg_iteration"]["ms"], r["avg_kernel_ms"], d["config"]["workload"][:110]))
PY
}
run cg_bj_216        --iters 100
run cg_none_216      --iters 100 --precond none
run cg_bj4_216       --iters 100 --block-size 4
run cg_isai_216      --iters 100 --precond ISAI
run bicg_bj_128a     --iters 100 --solver GKOBiCGStab --asym --edge 128
run bicg_gisai_128a  --iters 100 --solver GKOBiCGStab --asym --edge 128 --precond GISAI
run bicg_bj_216a     --iters 50 --solver GKOBiCGStab --asym
run gmres30_bj_216   --iters 60 --solver GKOGMRES --krylov-dim 30
run gmres30_bj_368   --iters 60 --solver GKOGMRES --krylov-dim 30 --edge 368
run gmres30_bj_368e  --iters 60 --solver GKOGMRES --krylov-dim 30 --edge 368 --format Ell
run cg_bj_368        --iters 50 --edge 368
