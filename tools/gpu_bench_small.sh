#!/bin/bash
# Small-size (launch/latency-bound) measurements: configs[0] cavity 64^3 and neighbours
mkdir -p gpurun_out
run() {
  TAG=$1; shift
  python bench.py --steps 5 --warmup 2 --cpu-iters 0 --no-general-legs $SMALL_EXTRA "$@" > gpurun_out/small_$TAG.json 2> gpurun_out/small_$TAG.err || { echo "$TAG FAILED"; tail -3 gpurun_out/small_$TAG.err; return; }
  python - "$TAG" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/small_{sys.argv[1]}.json"))
r=d["roofline"]
print("%-20s turns/s=%9.1f  us/turn=%8.2f  spmv_us=%7.2f" % (sys.argv[1], d["value"], 1e3*d["cg_iteration"]["ms"], 1e3*r["avg_kernel_ms"]))
PY
}
run cg_bj_32   --iters 200 --edge 32
run cg_bj_64   --iters 200 --edge 64
run cg_bj_64np --iters 200 --edge 64 --no-profile
run cg_bj_100  --iters 200 --edge 100
run cg_bj_32np  --iters 200 --edge 32 --no-profile
run cg_bj_100np --iters 200 --edge 100 --no-profile
run cg_bj_128np --iters 200 --edge 128 --no-profile
run cg_bj_160np --iters 200 --edge 160 --no-profile
run cg_bj_128  --iters 200 --edge 128
run bicg_bj_64 --iters 200 --edge 64 --solver GKOBiCGStab --asym
run gmres_bj_64 --iters 200 --edge 64 --solver GKOGMRES --krylov-dim 30
