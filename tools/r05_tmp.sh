export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 600 python -m pytest tests/test_gpu_bicg_fold.py -q -x 2>&1 | tail -5
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r05e_pytest.txt
B="--steps 3 --warmup 2 --cpu-iters 0 --no-general-legs"
run() { T=$1; shift
  python bench.py $B "$@" > gpurun_out/r05d.json 2> gpurun_out/r05d.err || { echo "$T FAILED"; tail -3 gpurun_out/r05d.err; return; }
  python - "$T" <<'PY' | tee -a gpurun_out/r05e_ab.txt
import json,sys
d=json.load(open("gpurun_out/r05d.json")); r=d["roofline"]; t=d["solver_turn"]
print("%-40s turns/s=%8.1f us/turn=%6.1f spmv_us=%5.1f frac %.3f turn frac %.3f %s" % (sys.argv[1], d["value"], 1e3*t["ms"], 1e3*r["avg_kernel_ms"], r["frac"], t["frac_of_peak"], r["kernel"]))
PY
}
for rep in 1 2; do
for V in "" "--prop streamTurnSet=0"; do
run "c3_bicg_isai_128s $V"   --iters 100 --edge 128 --shuffle 65536 --solver GKOBiCGStab --asym --precond ISAI $V
run "bicg_bj_128a $V"        --iters 100 --solver GKOBiCGStab --asym --edge 128 $V
run "bicg_gisai_128a $V"     --iters 100 --solver GKOBiCGStab --asym --edge 128 --precond GISAI $V
run "c5_gmres_csr_184s $V"   --iters 60 --edge 184 --shuffle 65536 --solver GKOGMRES --krylov-dim 30 $V
run "gmres30_bj_128 $V"      --iters 60 --edge 128 --solver GKOGMRES --krylov-dim 30 $V
run "cg_bj4_128s $V"         --iters 100 --edge 128 --shuffle 65536 --block-size 4 $V
run "cg_isai_136 $V"         --iters 100 --edge 136 --precond ISAI $V
done
for V in "" "--prop bicgFold=0"; do
run "bicg_bj_64a $V"   --iters 200 --solver GKOBiCGStab --asym --edge 64 $V
run "bicg_bj_32a $V"   --iters 200 --solver GKOBiCGStab --asym --edge 32 $V
run "bicg_gisai_64a $V"   --iters 200 --solver GKOBiCGStab --asym --edge 64 --precond GISAI $V
done
done
for R in 2 4; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$R --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $R --steps 3 --warmup 1 --edge 128 --iters 100 --cpu-iters 0 > gpurun_out/r05e_ranks$R.json 2> gpurun_out/r05e_ranks$R.err; echo "ranks $R rc=$?"; tail -2 gpurun_out/r05e_ranks$R.err | cut -c1-300
python - $R <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/r05e_ranks{sys.argv[1]}.json")); t=d["config"]["transport"]
    print(d["value"], d["config"]["parallelism"]); print(json.dumps(t["rungs"])[:1500]); print(json.dumps(t["wait_us"])[:800])
except Exception as e: print("no json", e)
PY
done
