#!/bin/bash
# First GPU pass: smoke (with and without torch's HIP runtime loaded first), parity tests, small bench.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_notorch.log 2>&1; echo "smoke(no torch) rc=$?"
python -c "import torch; torch.cuda.init(); import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_torch.log 2>&1; echo "smoke(torch first) rc=$?"
tail -3 gpurun_out/smoke_notorch.log gpurun_out/smoke_torch.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --edge 64 --steps 3 --warmup 1 --iters 50 > gpurun_out/bench_n64.json 2> gpurun_out/bench_n64.err; echo "bench64 rc=$?"
cat gpurun_out/bench_n64.json; tail -5 gpurun_out/bench_n64.err
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_n216.json 2> gpurun_out/bench_n216.err; echo "bench216 rc=$?"
cat gpurun_out/bench_n216.json; tail -5 gpurun_out/bench_n216.err
