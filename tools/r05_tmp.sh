export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r05f_pytest.txt
B="--steps 3 --warmup 2 --cpu-iters 0 --no-general-legs"
run() { T=$1; shift
  python bench.py $B "$@" > gpurun_out/r05d.json 2> gpurun_out/r05d.err || { echo "$T FAILED"; tail -3 gpurun_out/r05d.err; return; }
  python - "$T" <<'PY' | tee -a gpurun_out/r05f_ab.txt
import json,sys
d=json.load(open("gpurun_out/r05d.json")); r=d["roofline"]; t=d["solver_turn"]
print("%-40s turns/s=%8.1f us/turn=%6.1f spmv_us=%5.1f frac %.3f csr-eq %.3f turn frac %.3f %s %s first set_matrix %.2f" % (sys.argv[1], d["value"], 1e3*t["ms"], 1e3*r["avg_kernel_ms"], r["frac"], r["csr_equivalent_frac"], t["frac_of_peak"], r["kernel"], d["config"].get("numbering"), d["boundary"]["first_set_matrix_s"]))
PY
}
for rep in 1 2; do
run "vor3m centres"  --voronoi 3000000 --iters 100
run "vor3m rcm"      --voronoi 3000000 --iters 100 --no-centres
run "vor1m centres"  --voronoi 1000000 --iters 100
run "vor1m rcm"      --voronoi 1000000 --iters 100 --no-centres
for V in "" "--prop gmresFold=0"; do
run "gmres30_bj_64 $V"   --iters 90 --solver GKOGMRES --krylov-dim 30 --edge 64 $V
run "gmres30_bj_32 $V"   --iters 90 --solver GKOGMRES --krylov-dim 30 --edge 32 $V
done
done
