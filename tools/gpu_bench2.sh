#!/bin/bash
# bench.py with 2 ranks on the single GPU: RCCL refuses the duplicate device, so this exercises the
# agreed fallback to the host-buffer transport and the whole N>1 code path of bench.py.
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --edge 64 --iters 50 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "rc=$?"
cat gpurun_out/bench_2rank.json | cut -c1-900
grep -v "amdgpu.ids" gpurun_out/bench_2rank.err | tail -5
