#!/bin/bash
# Can RCCL run 2 ranks on the one GPU of this box?  (coverage of the send/recv + all-reduce path)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29511 tests/dist_worker.py --mode gpu-rccl --shape 12,12,12 --procs 1,1,2 > gpurun_out/rccl2.log 2>&1
echo "rc=$?"
grep -E "ok|rror|NCCL|Duplicate|invalid" gpurun_out/rccl2.log | head -20
tail -5 gpurun_out/rccl2.log
