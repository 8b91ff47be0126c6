export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
run() { echo "== $*"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29577 tests/dist_worker.py "$@" 2>&1 | grep -E " ok|Mismatched|Error|error" | sort | uniq -c | head -12; }
run --mode gpu-peer --shape 64,64,64 --procs 2,2,2 --max-iter 10
run --mode gpu-peer --shape 128,128,128 --procs 2,2,2 --max-iter 10
run --mode gpu-peer --shape 192,192,192 --procs 2,2,2 --max-iter 10
run --mode gpu-peer --shape 192,192,192 --procs 2,2,2 --max-iter 10 --no-global 1
run --mode gpu-host --shape 272,272,272 --procs 2,2,2 --max-iter 10 --no-global 1
run --mode gpu-peer --shape 272,272,272 --procs 2,2,2 --max-iter 10 --no-global 1 --halo-fused 0
