#!/bin/bash
# round 2 evidence pass: benches (default with CPU baselines, shuffled), small sizes, solver configs,
# 2-rank bench with the self-check, rocprofv3 kernel stats + PMC passes (default and shuffled)
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 1 > gpurun_out/r02g_default.json 2> gpurun_out/r02g_default.err; echo "bench rc=$?"
for W in 512 4096 65536; do
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle $W > gpurun_out/r02g_shuffle_$W.json 2> gpurun_out/r02g_shuffle_$W.err || tail -3 gpurun_out/r02g_shuffle_$W.err
done
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle 65536 --renumber off > gpurun_out/r02g_shuffle_65536_off.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --no-compress > gpurun_out/r02g_nocompress.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --full-storage > gpurun_out/r02g_fullstorage.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02g_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-36s %7.1f it/s layout=%-4s renumbered=%-5s spmv %6.1f us frac %.3f moved_frac %.3f first set_matrix %.2f s" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], d["boundary"]["first_set_matrix_s"]))
    if "cpu_baseline" in d: print("   ", d["cpu_baseline"], "\n   ", d["cpu_baseline_omp"])
PY
bash tools/gpu_bench_small.sh 2>&1 | tee gpurun_out/r02_small.txt
bash tools/gpu_bench_configs.sh r02
for PEER in 1 0; do
OGL_BENCH_PEER=$PEER timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --edge 128 --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/r02g_2rank_peer$PEER.json 2> gpurun_out/r02g_2rank_peer$PEER.err; echo "2-rank peer=$PEER rc=$?"
python - $PEER <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/r02g_2rank_peer{sys.argv[1]}.json"))
    print(d["value"], d["config"]["parallelism"]); print(d["config"]["transport"]); print(d["config"]["selfcheck"])
except Exception as e:
    print("unreadable", e); print(open(f"gpurun_out/r02g_2rank_peer{sys.argv[1]}.err").read()[-1500:])
PY
done
bash tools/gpu_profile.sh r02 > gpurun_out/prof_r02.log 2>&1; tail -3 gpurun_out/prof_r02.log | cut -c1-200
bash tools/gpu_profile.sh r02_shuffle --shuffle 65536 > gpurun_out/prof_r02_shuffle.log 2>&1
bash tools/gpu_pmc.sh r02 > gpurun_out/pmc_r02.log 2>&1; tail -3 gpurun_out/pmc_r02.log | cut -c1-200
bash tools/gpu_pmc.sh r02_shuffle --shuffle 65536 > gpurun_out/pmc_r02_shuffle.log 2>&1
bash tools/gpu_markers.sh > gpurun_out/markers_r02.log 2>&1
