#!/bin/bash
# round 2, second GPU pass: distributed + renumber tests, parity deviation, 2-rank bench with the
# self-check (ranks share the one GPU of the box), default bench with the CPU baselines
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_distributed.py tests/test_gpu_renumber.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_gpu_b.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu_b.log | cut -c1-300
timeout 900 python tools/parity_deviation.py > gpurun_out/r02_parity_deviation.txt 2> gpurun_out/r02_parity_deviation.err; echo "deviation rc=$?"
grep -v "per check" gpurun_out/r02_parity_deviation.txt | head -80
for PEER in 1 0; do
OGL_BENCH_PEER=$PEER timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --edge 100 --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/r02_2rank_peer$PEER.json 2> gpurun_out/r02_2rank_peer$PEER.err; echo "2-rank peer=$PEER rc=$?"
python - $PEER <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/r02_2rank_peer{sys.argv[1]}.json"))
    print(d["value"], d["config"]["parallelism"]); print(d["config"]["transport"]); print(d["config"]["selfcheck"])
except Exception as e:
    print("unreadable", e); print(open(f"gpurun_out/r02_2rank_peer{sys.argv[1]}.err").read()[-1500:])
PY
done
python bench.py --steps 5 --warmup 1 > gpurun_out/r02_default_cpu.json 2> gpurun_out/r02_default_cpu.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_default_cpu.json"))
print(d["value"], d["roofline"]["frac"]); print(d.get("cpu_baseline")); print(d.get("cpu_baseline_omp"))
PY
