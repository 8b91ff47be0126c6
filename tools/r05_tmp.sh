export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r05g_pytest.txt
B="--steps 3 --warmup 2 --cpu-iters 0 --no-general-legs"
run() { T=$1; shift
  python bench.py $B "$@" > gpurun_out/r05d.json 2> gpurun_out/r05d.err || { echo "$T FAILED"; tail -3 gpurun_out/r05d.err; return; }
  python - "$T" <<'PY' | tee -a gpurun_out/r05g_ab.txt
import json,sys
d=json.load(open("gpurun_out/r05d.json")); r=d["roofline"]; t=d["solver_turn"]
print("%-40s turns/s=%8.1f us/turn=%6.1f spmv_us=%5.1f frac %.3f csr-eq %.3f turn frac %.3f %s curve=%s first set_matrix %.2f" % (sys.argv[1], d["value"], 1e3*t["ms"], 1e3*r["avg_kernel_ms"], r["frac"], r["csr_equivalent_frac"], t["frac_of_peak"], r["kernel"], d["config"].get("numbering",{}).get("along_hilbert_curve"), d["boundary"]["first_set_matrix_s"]))
PY
}
for rep in 1 2; do
run "vor3m centres"  --voronoi 3000000 --iters 100
run "vor3m rcm"      --voronoi 3000000 --iters 100 --no-centres
run "vor1m centres"  --voronoi 1000000 --iters 100
run "vor1m rcm"      --voronoi 1000000 --iters 100 --no-centres
run "216 nocompress"             --no-compress
run "216 nocompress band46656"   --no-compress --prop spmvBandRows=46656
run "216 fullstorage"            --full-storage
run "216 fullstorage band46656"  --full-storage --prop spmvBandRows=46656
done
run "config3 N=1" --config 3 --iters 50
run "config4 N=1 csr" --config 4 --iters 30
run "config4 N=1 ell" --config 4 --iters 30 --format Ell
for C in 3 4; do
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 8 --config $C --steps 2 --warmup 1 --iters 40 --cpu-iters 0 --rung-timeout 400 > gpurun_out/r05g_config${C}_8ranks.json 2> gpurun_out/r05g_config${C}_8ranks.err; echo "config $C 8 ranks rc=$?"; tail -2 gpurun_out/r05g_config${C}_8ranks.err | cut -c1-300
python - $C <<'PY' | tee -a gpurun_out/r05g_ranks_configs.txt
import json,sys
try:
    d=json.load(open(f"gpurun_out/r05g_config{sys.argv[1]}_8ranks.json")); t=d["config"]["transport"]
    print("config", sys.argv[1], "value", d["value"], d["unit"], d["scaling"], "|", d["config"]["workload"][:160])
    print("   ", d["config"]["parallelism"], "| selfcheck", (d["config"]["selfcheck"] or {}).get("ok"), "| us/turn", 1e3*d["solver_turn"]["ms"])
    for r in t["rungs"]: print("    rung", {k:v for k,v in r.items() if k not in ("wait_us","transport")})
    for w in t["wait_us"]: print("    wait", w)
except Exception as e: print("no json", e)
PY
done
