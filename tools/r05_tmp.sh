export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -k "distributed or components or cpp_host or renumber or parity" 2>&1 | tail -6 | tee gpurun_out/r05i_pytest.txt
B="--steps 3 --warmup 2 --cpu-iters 0 --no-general-legs"
run() { T=$1; shift
  python bench.py $B "$@" > gpurun_out/r05d.json 2> gpurun_out/r05d.err || { echo "$T FAILED"; tail -3 gpurun_out/r05d.err; return; }
  python - "$T" <<'PY' | tee -a gpurun_out/r05i_ab.txt
import json,sys
d=json.load(open("gpurun_out/r05d.json")); r=d["roofline"]; t=d["solver_turn"]
print("%-44s turns/s=%8.1f us/turn=%6.1f spmv_us=%5.1f frac %.3f turn frac %.3f %s | components %s" % (sys.argv[1], d["value"], 1e3*t["ms"], 1e3*r["avg_kernel_ms"], r["frac"], t["frac_of_peak"], r["kernel"], (d["boundary"].get("momentum_components") or {}).get("refresh_ms")))
PY
}
for rep in 1 2; do
run "cg_bj4_128s one-pass (default)"        --iters 100 --edge 128 --shuffle 65536 --block-size 4
run "cg_bj4_128s staged (bjFusedPerm 0)"    --iters 100 --edge 128 --shuffle 65536 --block-size 4 --prop bjFusedPerm=0
run "cg_bj4_128s backend's own blocks"      --iters 100 --edge 128 --shuffle 65536 --block-size 4 --prop precondCallerNumbering=0
run "cg_bj8_128s one-pass (default)"        --iters 100 --edge 128 --shuffle 65536 --block-size 8
run "cg_bj8_128s staged (bjFusedPerm 0)"    --iters 100 --edge 128 --shuffle 65536 --block-size 8 --prop bjFusedPerm=0
run "cg_bj_216 (momentum components)"       --iters 100
done
bash tools/gpu_pass.sh r05i prof:--iters+100+--edge+128+--shuffle+65536+--solver+GKOBiCGStab+--asym+--precond+ISAI
