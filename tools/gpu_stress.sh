#!/bin/bash
# Soak of the peer mesh: R ranks sharing the GPU, many solves, every one compared bit for bit.
#   gpu_stress.sh RANKS SOLVES SEED
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
R=${1:-4}; S=${2:-200}; SEED=${3:-1}
mkdir -p gpurun_out
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$R --master-addr 127.0.0.1 --master-port 29533 tests/peer_stress_worker.py --solves $S --seed $SEED > gpurun_out/stress_$R.log 2>&1
echo "rc=$?"
grep -E "solves ok|Error|rror:|Mismatch" gpurun_out/stress_$R.log | head -12
