export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 1200 python -m pytest tests/test_gpu_proxy_meshes.py tests/test_gpu_sell.py -m gpu -q -x 2>&1 | tail -6
B="--steps 3 --warmup 2 --cpu-iters 0 --no-general-legs"
run() { T=$1; shift
  python bench.py $B "$@" > gpurun_out/r05d.json 2> gpurun_out/r05d.err || { echo "$T FAILED"; tail -3 gpurun_out/r05d.err; return; }
  python - "$T" <<'PY' | tee -a gpurun_out/r05k_ab.txt
import json,sys
d=json.load(open("gpurun_out/r05d.json")); r=d["roofline"]; t=d["solver_turn"]
print("%-40s turns/s=%8.1f us/turn=%6.1f spmv_us=%5.1f frac %.3f csr-eq %.3f turn frac %.3f %s curve=%s first set_matrix %.2f" % (sys.argv[1], d["value"], 1e3*t["ms"], 1e3*r["avg_kernel_ms"], r["frac"], r["csr_equivalent_frac"], t["frac_of_peak"], r["kernel"], d["config"].get("numbering",{}).get("along_hilbert_curve"), d["boundary"]["first_set_matrix_s"]))
PY
}
for rep in 1 2; do
run "vor3m centres"  --voronoi 3000000 --iters 100
run "vor3m rcm"      --voronoi 3000000 --iters 100 --no-centres
C3="--iters 100 --edge 128 --shuffle 65536 --solver GKOBiCGStab --asym --precond ISAI"
run "c3 default" $C3
run "c3 band 16384" $C3 --prop spmvBandRows=16384
run "c3 band 32768" $C3 --prop spmvBandRows=32768
run "c3 centres" $C3 --centres
run "c3_cg default" --iters 100 --edge 128 --shuffle 65536
run "c3_cg band 32768" --iters 100 --edge 128 --shuffle 65536 --prop spmvBandRows=32768
run "c3_cg centres" --iters 100 --edge 128 --shuffle 65536 --centres
run "216s default" --iters 100 --shuffle 65536
run "216s band 46656" --iters 100 --shuffle 65536 --prop spmvBandRows=46656
run "216s band 93312" --iters 100 --shuffle 65536 --prop spmvBandRows=93312
run "216s centres" --iters 100 --shuffle 65536 --centres
done
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r05k_pytest.txt
