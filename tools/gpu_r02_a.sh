#!/bin/bash
# round 2, first GPU pass: the GPU test-suite, then the library's own renumbering on shuffled boxes
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
python bench.py --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/r02_default.json 2> gpurun_out/r02_default.err || tail -5 gpurun_out/r02_default.err
for W in 512 4096 65536; do
  for R in auto off; do
    python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle $W --renumber $R > gpurun_out/r02_shuffle_${W}_$R.json 2> gpurun_out/r02_shuffle_${W}_$R.err || { echo "W=$W $R FAILED"; tail -3 gpurun_out/r02_shuffle_${W}_$R.err; continue; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-44s %7.1f it/s layout=%-4s renumbered=%-5s spmv %6.1f us frac %.3f sectors %.3f->%.3f first set_matrix %.2f s" % (
        f.split("/")[-1], d["value"], r["layout"], c["renumbered"], 1e3*r["avg_kernel_ms"], r["frac"],
        c["gather_sectors_per_entry"]["as_given"], c["gather_sectors_per_entry"]["in_use"], d["boundary"]["first_set_matrix_s"]))
PY
