# round 4 pass i: ISAI under renumber A/B (structures in the caller's / the backend's numbering), staging helpers spread over L3 domains A/B
export HSA_ENABLE_IPC_MODE_LEGACY=0
for CN in 1 0 1 0; do
python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --solver GKOBiCGStab --asym --edge 128 --shuffle 65536 --precond ISAI --prop precondCallerNumbering=$CN > gpurun_out/r04i_isai_cn$CN.json 2> gpurun_out/r04i_isai_cn$CN.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04i_isai_cn$CN.json")); print("bicg+ISAI 128^3 shuffled, precondCallerNumbering=$CN", d["value"], "turns/s, turn us", 1e3*d["solver_turn"]["ms"], "spmv us", 1e3*d["roofline"]["avg_kernel_ms"])
PY
python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size 4 --prop precondCallerNumbering=$CN > gpurun_out/r04i_bj4_cn$CN.json 2> gpurun_out/r04i_bj4_cn$CN.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04i_bj4_cn$CN.json")); print("cg+BJ4 128^3 shuffled, precondCallerNumbering=$CN", d["value"], "turns/s, turn us", 1e3*d["solver_turn"]["ms"])
PY
done 2>&1 | tee gpurun_out/r04i_precond_numbering.txt
for SP in 1 0 1 0; do
OGL_STAGE_SPREAD=$SP python bench.py --steps 3 --warmup 1 --cpu-iters 0 --no-general-legs > gpurun_out/r04i_bench_spread$SP.json 2> gpurun_out/r04i_bench_spread$SP.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04i_bench_spread$SP.json")); print("OGL_STAGE_SPREAD=$SP", d["value"], d["boundary"])
PY
done 2>&1 | tee gpurun_out/r04i_stage_spread.txt
lscpu | grep -E "Model name|Socket|L3|NUMA node\(s\)|Core" ; cat /sys/fs/cgroup/cpu.max; nproc
