#!/bin/bash
mkdir -p gpurun_out
for Q in 1 2 3; do
OGL_SORT_QUANT=$Q python bench.py --steps 3 --warmup 1 --cpu-iters 0 --drop-faces 0.3 > gpurun_out/r02i_drop_q$Q.json 2> gpurun_out/r02i_drop_q$Q.err || tail -3 gpurun_out/r02i_drop_q$Q.err
done
python bench.py --steps 3 --warmup 1 --cpu-iters 0 --drop-faces 0.3 --renumber off > gpurun_out/r02i_drop_off.json 2> gpurun_out/r02i_drop_off.err
OGL_SORT_QUANT=1 python bench.py --steps 3 --warmup 1 --cpu-iters 0 --drop-faces 0.3 --shuffle 65536 > gpurun_out/r02i_drop_shuffle_q1.json 2> /dev/null
OGL_SORT_QUANT=2 python bench.py --steps 3 --warmup 1 --cpu-iters 0 --drop-faces 0.3 --shuffle 65536 > gpurun_out/r02i_drop_shuffle_q2.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02i_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; c=d["config"]
    print("%-30s %7.1f it/s layout=%-4s renumbered=%-5s sorted=%-5s spmv %6.1f us frac %.3f moved_frac %.3f sectors %.3f nnz %d" % (
        f.split("/")[-1][5:-5], d["value"], r["layout"], c["renumbered"], c.get("rows_sorted_by_length"), 1e3*r["avg_kernel_ms"], r["frac"], r["moved_frac"], c["gather_sectors_per_entry"]["in_use"], c["nnz_per_gpu"]))
PY
