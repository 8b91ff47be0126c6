#!/bin/bash
# round 2, fifth GPU pass: full suite, benches (default x2, shuffled), small sizes, solver configs,
# rocprofv3 kernel stats + PMC passes of the default bench
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu.log | cut -c1-300
python bench.py --steps 5 --warmup 1 > gpurun_out/r02e_default.json 2> gpurun_out/r02e_default.err; echo "bench rc=$?"
python bench.py --steps 5 --warmup 1 --cpu-iters 0 > gpurun_out/r02e_default2.json 2> gpurun_out/r02e_default2.err
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle 65536 > gpurun_out/r02e_shuffle_65536.json 2> gpurun_out/r02e_shuffle_65536.err || tail -3 gpurun_out/r02e_shuffle_65536.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02e_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-40s %7.1f it/s layout=%-4s renumbered=%-5s spmv %6.1f us frac %.3f moved_frac %.3f first set_matrix %.2f s" % (
        f.split("/")[-1], d["value"], r["layout"], c["renumbered"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], d["boundary"]["first_set_matrix_s"]))
    if "cpu_baseline" in d: print("   ", d["cpu_baseline"], "\n   ", d["cpu_baseline_omp"])
PY
bash tools/gpu_bench_small.sh 2>&1 | tee gpurun_out/r02_small.txt
bash tools/gpu_bench_configs.sh r02
bash tools/gpu_profile.sh r02 > gpurun_out/prof_r02.log 2>&1; tail -4 gpurun_out/prof_r02.log | cut -c1-200
bash tools/gpu_pmc.sh r02 > gpurun_out/pmc_r02.log 2>&1; tail -5 gpurun_out/pmc_r02.log
bash tools/gpu_pmc.sh r02_shuffle --shuffle 65536 > gpurun_out/pmc_r02_shuffle.log 2>&1; tail -5 gpurun_out/pmc_r02_shuffle.log
