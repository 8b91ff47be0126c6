#!/bin/bash
# CSR-stream kernel on an unstructured-like numbering (bench.py --shuffle W): the compressed layouts do
# not qualify, layout = csr
mkdir -p gpurun_out
for W in 512 4096 65536; do
  python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle $W > gpurun_out/shuffle_$W.json 2> gpurun_out/shuffle_$W.err || { echo "W=$W FAILED"; tail -3 gpurun_out/shuffle_$W.err; continue; }
  python - $W <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/shuffle_{sys.argv[1]}.json")); r=d["roofline"]
print("shuffle window %6s: %7.1f it/s  layout=%s  spmv %.1f us  frac %.3f  first set_matrix %.2f s" % (sys.argv[1], d["value"], r["layout"], 1e3*r["avg_kernel_ms"], r["frac"], d["boundary"]["first_set_matrix_s"]))
PY
done
# the same shuffled cases after reverse Cuthill-McKee (what renumberMesh does)
for W in 4096 65536; do
  python bench.py --steps 3 --warmup 1 --cpu-iters 0 --shuffle $W --rcm > gpurun_out/shuffle_rcm_$W.json 2> gpurun_out/shuffle_rcm_$W.err || { echo "W=$W rcm FAILED"; tail -3 gpurun_out/shuffle_rcm_$W.err; continue; }
  python - $W <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/shuffle_rcm_{sys.argv[1]}.json")); r=d["roofline"]
print("shuffle window %6s + RCM: %7.1f it/s  layout=%s  spmv %.1f us  frac %.3f" % (sys.argv[1], d["value"], r["layout"], 1e3*r["avg_kernel_ms"], r["frac"]))
PY
done
