#!/bin/bash
# round 2, fourth GPU pass: grid-barrier microbenchmark, full GPU suite, shuffled bench, default bench
# with CPU baselines, rocprofv3 kernel stats of the default bench and of the shuffled one
mkdir -p gpurun_out
timeout 300 tools/bin/coop_tune 200 > gpurun_out/r02_coop_tune.txt 2>&1; echo "coop rc=$?"; cat gpurun_out/r02_coop_tune.txt
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle 65536 > gpurun_out/r02d_shuffle_65536.json 2> gpurun_out/r02d_shuffle_65536.err || tail -3 gpurun_out/r02d_shuffle_65536.err
python bench.py --steps 5 --warmup 1 > gpurun_out/r02d_default.json 2> gpurun_out/r02d_default.err; echo "bench rc=$?"
bash tools/gpu_profile.sh r02 > gpurun_out/prof_r02.log 2>&1; tail -25 gpurun_out/prof_r02.log | cut -c1-200
bash tools/gpu_profile.sh r02_shuffle --shuffle 65536 > gpurun_out/prof_r02_shuffle.log 2>&1; tail -12 gpurun_out/prof_r02_shuffle.log | cut -c1-200
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02d_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-40s %7.1f it/s layout=%-4s renumbered=%-5s spmv %6.1f us frac %.3f moved_frac %.3f first set_matrix %.2f s" % (
        f.split("/")[-1], d["value"], r["layout"], c["renumbered"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], d["boundary"]["first_set_matrix_s"]))
    if "cpu_baseline" in d: print("   ", d["cpu_baseline"], "\n   ", d["cpu_baseline_omp"])
PY
