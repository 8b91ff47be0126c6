#!/bin/bash
# round 2, third GPU pass: full GPU suite, delta16 SELL on the shuffled boxes, default bench + CPU baselines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
for W in 512 65536; do
  python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle $W > gpurun_out/r02c_shuffle_${W}.json 2> gpurun_out/r02c_shuffle_${W}.err || { echo "W=$W FAILED"; tail -3 gpurun_out/r02c_shuffle_${W}.err; }
done
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle 65536 --no-compress > gpurun_out/r02c_shuffle_65536_nocompress.json 2> gpurun_out/r02c_shuffle_65536_nocompress.err
python bench.py --steps 5 --warmup 1 > gpurun_out/r02c_default.json 2> gpurun_out/r02c_default.err; echo "bench rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02c_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-40s %7.1f it/s layout=%-4s renumbered=%-5s spmv %6.1f us frac %.3f moved_frac %.3f first set_matrix %.2f s" % (
        f.split("/")[-1], d["value"], r["layout"], c["renumbered"], 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], d["boundary"]["first_set_matrix_s"]))
    if "cpu_baseline" in d: print("   ", d["cpu_baseline"], "\n   ", d["cpu_baseline_omp"])
PY
